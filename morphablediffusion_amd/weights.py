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


def seeded_tensor(key: str, shape: Tuple[int, ...], seed: int = 0, style: str = "init") -> torch.Tensor:
    """style "init": the reference's default-initialisation statistics.  style "trained": statistics of a trained
    checkpoint -- norm gains far from 1 (log-normal, sigma 0.4), biases of a few tenths, heavy-tailed (Student-t like)
    conv / linear weights, and the reference's zero-initialised output projections (``proj_out``, ``to_out``, the
    second conv of each ResBlock / DepthTransformer) with 2x the default norm -- so that parity is shown on a second,
    harder weight distribution and not only on one seed of one."""
    g = _gen(key, seed)
    if style == "trained":
        return _trained_tensor(key, shape, g)
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


_LOUD = ("proj_out.weight", "to_out.0.weight", "out_layers.3.weight", "proj_out.5.weight", ".out.2.weight")


def _trained_tensor(key: str, shape: Tuple[int, ...], g: torch.Generator) -> torch.Tensor:
    if key.endswith("running_var"):
        return 0.25 + 1.5 * torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.3 * torch.randn(shape, generator=g)
    if len(shape) == 1:
        if key.endswith(".weight"):  # norm gain: log-normal around 1
            return torch.exp(0.4 * torch.randn(shape, generator=g))
        return 0.2 * torch.randn(shape, generator=g)
    n = 1
    for s in shape:
        n *= s
    fan_in = n // shape[0]
    # heavy tails: normal scaled by a per-element inverse-chi factor (Student-t, 5 dof), same variance as "init"
    z = torch.randn(shape, generator=g)
    chi = (torch.randn((5,) + tuple(shape), generator=g) ** 2).mean(0).sqrt()
    w = z / chi * (3.0 / 5.0) ** 0.5 * (1.0 / (3.0 * fan_in)) ** 0.5
    if key.endswith(_LOUD):
        w = w * 2.0
    return w


def seeded_state_dict(manifest: Dict[str, Tuple[int, ...]], seed: int = 0, style: str = "init") -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, tuple(v), seed, style) for k, v in manifest.items()}
