"""GPU: error behaviour of the C ABI (non-zero status + mvd_last_error -> MvdError) and recovery after a failed call.
The reference raises Python exceptions for these cases (missing state_dict keys: load_state_dict; out of memory:
torch's allocator); a failed call must leave the context usable."""
import pytest
import torch

from morphablediffusion_amd import lib as L
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu


def _unet_engine(workspace_gb):
    from morphablediffusion_amd.engine import Engine
    cfg = gi.SMALL_UNET
    e = Engine(cfg, VolumeConfig(), workspace_gb=workspace_gb)
    e.load_state_dict(gi.unet_weights(cfg))
    return e, cfg


def _fwd(e, cfg, Bv, seed=3):
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=Bv, seed=seed, zero_uncond=False)
    return e.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), {k: v.cuda() for k, v in sd.items()})


def test_workspace_exhaustion_is_reported_and_does_not_leak():
    """A batch that does not fit the workspace fails with a message naming the remedy; the bump allocator is rolled
    back, so the same context then runs a smaller batch and gets the same numbers as a fresh context."""
    small, cfg = _unet_engine(0.12)
    ref, _ = _unet_engine(2.0)
    want = _fwd(ref, cfg, 1)
    before = _fwd(small, cfg, 1)
    assert torch.allclose(before, want, rtol=0, atol=1e-4 * want.abs().max().item())
    for _ in range(3):  # repeated failures must not eat the workspace either
        with pytest.raises(L.MvdError, match="workspace"):
            _fwd(small, cfg, 24)
    assert torch.equal(_fwd(small, cfg, 1), before)
    small.close()
    ref.close()


def test_missing_weight_is_named():
    from morphablediffusion_amd.engine import Engine
    cfg = gi.SMALL_UNET
    W = gi.unet_weights(cfg)
    victim = "model.diffusion_model.middle_block.0.in_layers.2.weight"
    del W[victim]
    e = Engine(cfg, VolumeConfig(), workspace_gb=0.5)
    with pytest.raises(L.MvdError, match=victim.replace(".", r"\.")):
        e.load_state_dict(W)
    e.close()


def test_calls_before_weights_and_without_optional_sections_fail_loudly():
    from morphablediffusion_amd.engine import Engine
    cfg = gi.SMALL_UNET
    e = Engine(cfg, VolumeConfig(), workspace_gb=0.5)
    with pytest.raises(L.MvdError, match="finalized"):
        _fwd(e, cfg, 1)
    e.load_state_dict(gi.unet_weights(cfg))
    with pytest.raises(L.MvdError):
        e.vae_decode(torch.zeros(1, 4, 32, 32))
    with pytest.raises(L.MvdError):
        e.clip_encode(torch.zeros(1, 3, 256, 256))
    with pytest.raises(L.MvdError):  # uploading after finalize is an error, not a silent no-op
        e.load_state_dict(gi.unet_weights(cfg))
    assert torch.isfinite(_fwd(e, cfg, 1)).all()
    e.close()


def test_unsupported_arguments_keep_the_reference_exceptions():
    e, cfg = _unet_engine(1.0)
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=1)
    with pytest.raises(NotImplementedError):  # more than one context token (the reference always passes one)
        e.unet_forward(x.cuda(), t.cuda(), torch.zeros(1, 2, 768).cuda(), {k: v.cuda() for k, v in sd.items()})
    sd.pop(8)
    with pytest.raises(KeyError):             # source_dict[res] lookup, attention.py:127-135
        e.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), {k: v.cuda() for k, v in sd.items()})
    e.close()
