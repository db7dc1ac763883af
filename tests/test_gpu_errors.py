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
    # C ABI: uploading into a finalized context is an error, not a silent no-op ...
    import ctypes as C
    w = torch.zeros(4)
    rc = e.lib.mvd_upload_weight(e._ctx, b"model.diffusion_model.out.2.bias", L.ptr(w), (C.c_int64 * 1)(4), 1, 0)
    assert rc != 0 and b"finalized" in e.lib.mvd_last_error()
    # ... while the Python surface reloads like nn.Module does (a fresh context behind the same Engine object)
    ref = _fwd(e, cfg, 1)
    inc = e.load_state_dict(gi.unet_weights(cfg))
    assert inc.unexpected_keys == [] and any(k.startswith("spatial_volume.") for k in inc.missing_keys)
    assert torch.equal(_fwd(e, cfg, 1), ref)
    e.close()


def test_set_mesh_is_transactional_and_sample_slots_keep_their_tables():
    """A failed mvd_set_mesh (voxel index outside out_sh) leaves the active mesh intact; slots switched with
    mvd_select_sample keep their own tables (B > 1 does not rebuild them every step)."""
    from morphablediffusion_amd import synthetic
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import VolumeConfig
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    e = Engine(ucfg, vcfg, workspace_gb=2.0)
    e.load_state_dict(gi.full_weights(ucfg, vcfg))
    b0 = synthetic.make_batch(N, "perspective", 500, mesh_seed=1)
    b1 = synthetic.make_batch(N, "perspective", 400, mesh_seed=2, radii=(0.2, 0.25, 0.27))
    x = torch.randn(N, 4, 32, 32, generator=torch.Generator().manual_seed(0)).cuda()
    te, ve = torch.zeros(256).cuda(), torch.zeros(N, 4).cuda()

    def vol(b, slot):
        e.select_sample(slot)
        return e.volume_from_fused(e.vertex_features(x, te, ve, torch.arange(N)))

    def upload(b, slot):
        e.select_sample(slot)
        e.set_mesh(b["vertices"][0], b["coord"][0], b["out_sh"][0], b["bounds"][0])
        e.set_cameras(b["target_K"][0], b["target_RT"][0])

    upload(b0, 0)
    v0 = vol(b0, 0)
    bad = b0["coord"][0].clone()
    bad[7, 1] = b0["out_sh"][0, 1] + 3
    with pytest.raises(L.MvdError, match="outside out_sh"):
        e.set_mesh(b0["vertices"][0], bad, b0["out_sh"][0], b0["bounds"][0])
    assert torch.equal(vol(b0, 0), v0)  # the old tables are still there and still consistent
    upload(b1, 1)
    v1 = vol(b1, 1)
    assert not torch.equal(v0, v1)
    assert torch.equal(vol(b0, 0), v0) and torch.equal(vol(b1, 1), v1)  # switching back and forth, no re-upload
    e.select_sample(5)  # an empty slot has no mesh
    with pytest.raises(L.MvdError, match="mvd_set_mesh"):
        e.vertex_features(x, te, ve, torch.arange(N))
    e.close()


def test_set_samples_equals_per_sample_upload_and_validates_the_whole_batch_first():
    """mvd_set_samples_async (a training step's new batch: rule books built on host threads, stream-ordered upload) gives the
    same tables as mvd_set_mesh / mvd_set_cameras per sample; one bad sample fails the call before ANY table is replaced."""
    from morphablediffusion_amd import synthetic
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import VolumeConfig
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    e = Engine(ucfg, vcfg, workspace_gb=2.0)
    e.load_state_dict(gi.full_weights(ucfg, vcfg))
    bs = [synthetic.make_batch(N, "perspective", 500 + 60 * i, mesh_seed=3 + i, radii=(0.2 + 0.01 * i, 0.25, 0.27)) for i in range(3)]
    x = torch.randn(N, 4, 32, 32, generator=torch.Generator().manual_seed(0)).cuda()
    te, ve = torch.zeros(256).cuda(), torch.zeros(N, 4).cuda()

    def vol(slot):
        e.select_sample(slot)
        return e.volume_from_fused(e.vertex_features(x, te, ve, torch.arange(N)))

    ref = []
    for i, b in enumerate(bs):  # one by one, synchronous entries
        e.select_sample(10 + i)
        e.set_mesh(b["vertices"][0], b["coord"][0], b["out_sh"][0], b["bounds"][0])
        e.set_cameras(b["target_K"][0], b["target_RT"][0])
        ref.append(vol(10 + i))
    args = lambda key: [b[key][0] for b in bs]
    e.set_samples([0, 1, 2], args("vertices"), args("coord"), args("out_sh"), args("bounds"), args("target_K"), args("target_RT"))
    for i in range(3):
        assert torch.equal(vol(i), ref[i])
    # a second upload into the same slots (other order: pools are reused, sizes differ), then a batch with one bad sample
    order = [2, 0, 1]
    pick = lambda key: [bs[j][key][0] for j in order]
    e.set_samples([0, 1, 2], pick("vertices"), pick("coord"), pick("out_sh"), pick("bounds"), pick("target_K"), pick("target_RT"))
    for i, j in enumerate(order):
        assert torch.equal(vol(i), ref[j])
    bad = args("coord")
    bad[2] = bad[2].clone()
    bad[2][5, 0] = bs[2]["out_sh"][0, 0] + 1
    with pytest.raises(L.MvdError, match="outside out_sh"):
        e.set_samples([0, 1, 2], args("vertices"), bad, args("out_sh"), args("bounds"), args("target_K"), args("target_RT"))
    for i, j in enumerate(order):  # nothing was replaced
        assert torch.equal(vol(i), ref[j])
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
