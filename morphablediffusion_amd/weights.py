"""Deterministic, name-keyed synthetic weights.

There is no network for checkpoints, and a full-size checkpoint (1 GB) cannot be committed as a fixture,
so goldens, tests and bench.py all use the same seeded initialisation, reproducible from (key, shape, seed)
alone.  Zero-initialised convs of the reference (zero_module at openaimodel.py:230,720, attention.py:71,
modules/attention.py:319) are given random values too: with them at zero every residual branch and the
final ``out`` conv would output exactly 0 and parity would be vacuous (SURVEY.md gotcha G5).
"""
import zlib
from typing import Dict, Tuple

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    g = _gen(key, seed)
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 1:
        if key.endswith(".weight"):  # norm gain
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)  # any bias
    n = 1
    for s in shape:
        n *= s
    fan_in = n // shape[0]
    # U(-1/sqrt(fan_in), 1/sqrt(fan_in)): the distribution of the reference's own default initialisation
    # (nn.Conv*/nn.Linear: kaiming_uniform_(a=sqrt(5)) -> bound = 1/sqrt(fan_in))
    return (torch.rand(shape, generator=g) * 2.0 - 1.0) * (1.0 / fan_in) ** 0.5


def seeded_state_dict(manifest: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, tuple(v), seed) for k, v in manifest.items()}
