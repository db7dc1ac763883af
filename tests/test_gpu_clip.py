"""GPU: CLIP image embedding through the C ABI against the goldens (transformers'
CLIPVisionModelWithProjection on the seeded weights, see oracle/clip_oracle.py) and against the CPU oracle on inputs
that exercise the bicubic resize in both directions.

Tolerance: north_star's 1e-3 (relative L2 and normalised max error).  Measured: 2.9e-4 (reduced width) / 4.3e-4
(ViT-L/14) relative L2 -- fp16 operands / fp32 accumulation through 24 pre-LN blocks, with the residual stream and
every LayerNorm input kept in fp32."""
REL_CLIP, MAX_CLIP = 1e-3, 1e-3
import os

import numpy as np
import pytest
import torch

from morphablediffusion_amd.spec import ClipConfig, UNetConfig, VolumeConfig, clip_manifest
from morphablediffusion_amd.weights import seeded_state_dict
from tests import golden_inputs as gi
from tests.test_gpu_model import compare

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
SMALL = dict(width=128, layers=2, heads=2, embed=64)


def _engine(cfg, workspace_gb=1.0):
    from morphablediffusion_amd.engine import Engine
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=workspace_gb)
    W = seeded_state_dict(clip_manifest(cfg), gi.WEIGHT_SEED)
    e.load_state_dict(W)
    return e, W


@pytest.mark.parametrize("name,kw", [("clip_small.npz", SMALL), ("clip_full.npz", {})])
def test_clip_encode_vs_golden(name, kw):
    g = np.load(os.path.join(G, name))
    cfg = ClipConfig(**kw)
    e, _ = _engine(cfg)
    gen = torch.Generator().manual_seed(47)
    x = torch.rand(int(g["B"]), 3, 256, 256, generator=gen) * 2.0 - 1.0
    out = e.clip_encode(x.cuda())
    assert out.shape == (int(g["B"]), 1, cfg.embed)
    compare(out, g, "embed", rel=REL_CLIP, mx=MAX_CLIP)
    e.close()


@pytest.mark.parametrize("H,W,B", [(224, 224, 1), (200, 320, 3), (512, 512, 2)])
def test_clip_encode_vs_oracle_resize(H, W, B):
    """Identity resize (224), mixed up / down sampling of a non-square image, 512^2 inputs (config 4), B > 1."""
    from oracle import clip_oracle as CO
    cfg = ClipConfig(**SMALL)
    e, Wt = _engine(cfg)
    gen = torch.Generator().manual_seed(H * 7 + W)
    x = torch.rand(B, 3, H, W, generator=gen) * 2.0 - 1.0
    want = CO.encode(Wt, cfg, x)
    got = e.clip_encode(x.cuda()).cpu()
    rel = ((got - want).norm() / want.norm()).item()
    assert rel < REL_CLIP, rel
    # samples are independent: the first sample alone gives the same row
    solo = e.clip_encode(x[:1].cuda()).cpu()
    assert torch.allclose(solo[0], got[0], rtol=0, atol=2e-3 * want.abs().max().item())
    e.close()


def test_prepare_uses_engine_clip():
    """SyncMultiviewDiffusion.prepare (morphable_diffusion.py:473-489) takes the CLIP embedding from the engine when
    the clip_image_encoder.model.visual.* tensors are part of the loaded state_dict."""
    from oracle import clip_oracle as CO
    from tests.test_gpu_model import make_model
    from morphablediffusion_amd.spec import vae_encoder_manifest, VaeConfig
    cfg = ClipConfig(**SMALL)
    extra = seeded_state_dict(clip_manifest(cfg), gi.WEIGHT_SEED)
    extra.update(seeded_state_dict(vae_encoder_manifest(VaeConfig(ch=32)), gi.WEIGHT_SEED))
    model = make_model(gi.SMALL_UNET, VolumeConfig(), 4, workspace_gb=4.0, extra_weights=extra)
    gen = torch.Generator().manual_seed(5)
    img = torch.rand(2, 256, 256, 3, generator=gen) * 2.0 - 1.0
    batch = {"input_image": img.cuda(), "input_elevation": torch.zeros(2, 1).cuda()}
    _, clip_embed, info = model.prepare(batch)
    want = CO.encode(extra, cfg, img.permute(0, 3, 1, 2))
    assert clip_embed.shape == (2, 1, cfg.embed)
    assert ((clip_embed.cpu() - want).norm() / want.norm()).item() < REL_CLIP
    assert info["x"].shape == (2, 4, 32, 32)
