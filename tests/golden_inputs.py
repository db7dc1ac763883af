"""Seeded inputs shared by tools/make_goldens.py (reference side, build container) and the tests
(oracle / HIP side).  Only seeds and shapes live here -- no reference code, no data files."""
import torch

from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest, unet_manifest
from morphablediffusion_amd.weights import seeded_state_dict

SMALL_UNET = UNetConfig(model_channels=64)  # volume_dims stay (64,128,256,512): they are the frustum net's widths
FULL_UNET = UNetConfig()
WEIGHT_SEED = 7


def pack(t, limit=65536, target=16384):  # limit: largest tensor stored whole
    """Full tensor when small, else a strided sample + two checksums."""
    t = t.detach().float().contiguous()
    flat = t.flatten()
    if flat.numel() <= limit:
        return {"full": flat.numpy(), "shape": list(t.shape)}
    stride = max(1, flat.numel() // target) | 1  # odd: never aliases with the power-of-two tensor extents
    return {"sample": flat[::stride].numpy(), "stride": stride, "shape": list(t.shape),
            "sum": float(flat.double().sum()), "abssum": float(flat.double().abs().sum())}


def unpack_compare(t, g, prefix):
    """Returns (got, want) flat tensors for comparison + checksum pair (or None)."""
    import numpy as np
    t = t.detach().float().contiguous().flatten()
    if prefix + ".full" in g:
        return t, torch.from_numpy(np.asarray(g[prefix + ".full"])), None
    stride = int(g[prefix + ".stride"])
    sums = (float(t.double().sum()), float(g[prefix + ".sum"]), float(t.double().abs().sum()), float(g[prefix + ".abssum"]))
    return t[::stride], torch.from_numpy(np.asarray(g[prefix + ".sample"])), sums


def flatten_packs(d):
    out = {}
    for k, v in d.items():
        for kk, vv in v.items():
            out[f"{k}.{kk}"] = vv
    return out


def unet_inputs(cfg: UNetConfig, Bv=2, seed=11, zero_uncond=True):
    g = torch.Generator().manual_seed(seed)
    s = cfg.image_size
    x = torch.randn(Bv, cfg.in_channels, s, s, generator=g)
    t = torch.tensor([481] * Bv, dtype=torch.long)
    ctx = torch.randn(Bv, 1, cfg.context_dim, generator=g)
    D = 48 * s // 32
    sd = {}
    for lvl, c in enumerate(cfg.volume_dims):
        r = s >> lvl
        sd[r] = torch.randn(Bv, c, D >> lvl, r, r, generator=g)
    if zero_uncond and Bv > 1:
        for k in sd:
            sd[k][Bv // 2:] = 0
        ctx[Bv // 2:] = 0
    return x, t, ctx, sd


TRAINED_SEED = 23  # second weight set: trained-checkpoint-like statistics (weights.py, style "trained")


def unet_weights(cfg: UNetConfig, style="init"):
    return seeded_state_dict(unet_manifest(cfg), WEIGHT_SEED if style == "init" else TRAINED_SEED, style)


def full_weights(ucfg: UNetConfig, vcfg: VolumeConfig, style="init"):
    return seeded_state_dict(full_manifest(ucfg, vcfg), WEIGHT_SEED if style == "init" else TRAINED_SEED, style)
