"""GPU: first-stage decoder through the C ABI against the reference's AutoencoderKL.decode goldens
and against the CPU oracle on a batch.

Tolerance: relative L2 <= 3e-3, normalised max error <= 5e-3.  The decoder is ~30 convolutions in series with no
normalising residual structure around them; IDEAL fp16-operand / fp32-accumulate arithmetic applied to the reference
itself (operands of every conv and bmm rounded to fp16 on the CPU) already differs from the fp32 reference by 1.8e-3
(ch=128) / 1.9e-3 (ch=32) relative L2 on these seeded weights, and the HIP path measures 1.8e-3 / 2.0e-3, i.e. it
sits on that floor.  The north_star 1e-3 bound is stated for the UNet outputs, where the same arithmetic gives 8.5e-4.
In image terms the error is below half an 8-bit step."""
REL_VAE, MAX_VAE = 3e-3, 5e-3
import os

import numpy as np
import pytest
import torch

from morphablediffusion_amd.spec import UNetConfig, VaeConfig, VolumeConfig, vae_decoder_manifest, vae_encoder_manifest
from morphablediffusion_amd.weights import seeded_state_dict
from tests import golden_inputs as gi
from tests.test_gpu_model import compare

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _engine(cfg, workspace_gb, exact=False):
    from morphablediffusion_amd.engine import Engine
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=workspace_gb, vae_exact=exact)
    W = seeded_state_dict(vae_decoder_manifest(cfg), gi.WEIGHT_SEED)
    W.update(seeded_state_dict(vae_encoder_manifest(cfg), gi.WEIGHT_SEED))
    e.load_state_dict(W)
    return e, W


@pytest.mark.parametrize("name,ch,ws", [("vae_small.npz", 32, 2.0), ("vae_full.npz", 128, 6.0)])
def test_vae_decode_vs_golden(name, ch, ws):
    g = np.load(os.path.join(G, name))
    cfg = VaeConfig(ch=ch)
    e, _ = _engine(cfg, ws)
    gen = torch.Generator().manual_seed(31)
    z = torch.randn(int(g["B"]), cfg.embed_dim, 32, 32, generator=gen) * 4.0
    img = e.vae_decode(z.cuda())
    compare(img, g, "out", rel=REL_VAE, mx=MAX_VAE)
    # the adopted bar for the first stage: in IMAGE space -- what generate_face.py:244-252 writes --
    # the decoded views differ from the reference's by less than ONE 8-bit step everywhere (measured: 0.61 / 0.77 of a step
    # at the worst pixel) and by less than a tenth of a step on average, i.e. at most a last-bit flip after quantisation
    got, want, _ = gi.unpack_compare(img.float().cpu(), g, "out")
    to8 = lambda t: (torch.clamp(t, -1.0, 1.0) + 1.0) * 0.5 * 255.0
    d8 = (to8(got) - to8(want)).abs().max().item()
    m8 = (to8(got) - to8(want)).abs().mean().item()
    print(f"[parity] decoded image, 8-bit units: max |diff| = {d8:.3f}, mean |diff| = {m8:.4f}")
    assert d8 <= 1.0 and m8 <= 0.1
    e.close()


@pytest.mark.parametrize("name,ch,ws", [("vae_small.npz", 32, 3.0), ("vae_full.npz", 128, 8.0)])
def test_vae_encode_vs_golden(name, ch, ws):
    """Encoder + quant_conv (moments) against the reference's AutoencoderKL.encode(x).parameters."""
    g = np.load(os.path.join(G, name))
    cfg = VaeConfig(ch=ch)
    e, _ = _engine(cfg, ws)
    gen = torch.Generator().manual_seed(31)
    B = int(g["B"])
    torch.randn(B, cfg.embed_dim, 32, 32, generator=gen)  # the decoder golden's latent draw comes first
    x = torch.rand(B, 3, 256, 256, generator=gen) * 2.0 - 1.0
    compare(e.vae_encode_moments(x.cuda()), g, "moments", rel=REL_VAE, mx=MAX_VAE)
    e.close()


@pytest.mark.parametrize("name,ch,ws", [("vae_small.npz", 32, 4.0), ("vae_full.npz", 128, 16.0)])
def test_vae_exact_mode_meets_1e3(name, ch, ws):
    """f1 at north_star's stated tolerance: mvd_set_vae_precision(1) -- every convolution, the attention projections and both
    attention products in extended precision (fp16 hi + lo operand split, three products in fp32) -- reproduces the
    reference's fp32 decoder AND encoder to <= 1e-3 relative L2 (measured values printed; the default mode's 1.8-2.0e-3 is the
    fp16-operand floor).  Its cost next to the default mode is printed too."""
    import time
    g = np.load(os.path.join(G, name))
    cfg = VaeConfig(ch=ch)
    gen = torch.Generator().manual_seed(31)
    B = int(g["B"])
    z = torch.randn(B, cfg.embed_dim, 32, 32, generator=gen) * 4.0
    x = torch.rand(B, 3, 256, 256, generator=gen) * 2.0 - 1.0
    times = {}
    for exact in (True, False):
        e, _ = _engine(cfg, ws, exact=exact)
        img = e.vae_decode(z.cuda())
        mom = e.vae_encode_moments(x.cuda())
        if exact:
            compare(img, g, "out", rel=1e-3, mx=2e-3)
            compare(mom, g, "moments", rel=1e-3, mx=2e-3)
        zz = z.repeat(8, 1, 1, 1)[:8].cuda()  # 8 views at once: the regime of SyncMultiviewDiffusion.sample
        e.vae_decode(zz)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            e.vae_decode(zz)
        torch.cuda.synchronize()
        times[exact] = (time.perf_counter() - t0) / 3
        e.close()
    print(f"[cost] first-stage decode of 8 latents, ch={ch}: exact {1e3 * times[True]:.2f} ms vs default {1e3 * times[False]:.2f} ms "
          f"({times[True] / times[False]:.2f}x)")


def test_vae_decode_batch_vs_oracle():
    """A batch of 3 latents at reduced width against the CPU oracle, plus batch invariance (1 vs 3 at a time)."""
    from oracle import vae_oracle as V
    cfg = VaeConfig(ch=32)
    e, W = _engine(cfg, 3.0)
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(3, cfg.embed_dim, 32, 32, generator=gen) * 4.0
    want = V.decode(W, cfg, z)
    got = e.vae_decode(z.cuda()).cpu()
    rel = ((got - want).norm() / want.norm()).item()
    mx = ((got - want).abs().max() / want.abs().max()).item()
    print(f"[parity] vae decode batch vs oracle: relL2={rel:.2e} maxnorm={mx:.2e}")
    assert torch.isfinite(got).all() and rel <= REL_VAE and mx <= MAX_VAE
    one = torch.cat([e.vae_decode(z[i:i + 1].cuda()).cpu() for i in range(3)])
    d = ((one - got).norm() / got.norm()).item()
    print(f"[property] vae decode 1-at-a-time vs batched: relL2={d:.2e}")
    # not bit-identical: split-K / tile choices depend on the batch, i.e. the fp32 summation order; through ~30 layers of
    # fp16 operand rounding a 1e-7 perturbation saturates at the rounding-noise floor itself, so two batchings are two
    # realisations of the same noise (each ~2e-3 from the fp32 result, ~1e-3 from each other)
    assert d <= REL_VAE
    e.close()


def test_sample_end_to_end_small():
    """The whole drop-in surface once: build_batch -> SyncMultiviewDiffusion.sample (prepare with injected stand-ins for
    the frozen VAE encoder / CLIP, 6 DDIM steps of the HIP engine, batched HIP first-stage decode) -> images; the
    decoded result must equal decoding the sampler's latents by hand (same seed)."""
    import dataclasses
    from morphablediffusion_amd import batch as BT, synthetic
    from morphablediffusion_amd.model import SyncDDIMSampler
    from tests.test_gpu_model import make_model
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)

    class Posterior:
        def __init__(self, z): self.z = z
        def sample(self): return self.z
        def mode(self): return self.z

    class FakeVaeEncoder:  # frozen encoder stays host plumbing (north_star); deterministic stand-in
        def encode(self, x):
            return Posterior(torch.nn.functional.avg_pool2d(x, 8).mean(1, keepdim=True).repeat(1, 4, 1, 1))

    class FakeClip:
        def encode(self, x):
            g = torch.Generator().manual_seed(3)
            return torch.randn(x.shape[0], 1, 768, generator=g).to(x.device)

    m = make_model(ucfg, vcfg, N, workspace_gb=6.0,
                   extra_weights=seeded_state_dict(vae_decoder_manifest(VaeConfig(ch=32)), gi.WEIGHT_SEED))
    m.first_stage_model, m.clip_image_encoder = FakeVaeEncoder(), FakeClip()
    verts = synthetic.ellipsoid_mesh(600, 1)
    data = BT.build_batch(torch.zeros(256, 256, 3).uniform_(-1, 1), verts, num_views=N, device="cuda")
    sampler = SyncDDIMSampler(m, 6)
    torch.manual_seed(11)
    imgs = m.sample(sampler, data, 2.0, N)
    assert imgs.shape == (1, N, 3, 256, 256) and torch.isfinite(imgs).all()
    torch.manual_seed(11)
    _, clip, info = m.prepare(data)
    lat, _ = sampler.sample(info, clip, unconditional_scale=2.0, batch_view_num=N, batch=data)
    by_hand = m.engine.vae_decode(lat[0] / m.first_stage_scale_factor)
    assert torch.equal(by_hand, imgs[0])
    m.engine.close()
