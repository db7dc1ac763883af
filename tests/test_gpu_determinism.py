"""GPU: bit-reproducibility.  (Round 4: the non-determinism described below was root-caused -- a packed multiply with crossed
half selects in the frustum gather, see DESIGN.md section 4 -- and the last two tests pin the fix: the frustum volumes beside
halo convolutions of another stream, and the whole step under the launch order that used to fail.)
  The step runs the frustum network and the context halves of the DepthTransformers on a side
stream beside the UNet trunk (csrc/engine_unet.hip); every cross-stream dependency is an event, so repeating a step on the
same inputs must give the same bits.  (While that overlap was being built the step differed in ~5 % of the repetitions until
the fork became a two-way handshake -- this test is what caught it; see DESIGN.md.)  The second test repeats single kernels
while another stream saturates HBM with streaming and random-gather traffic: hand-counted LDS-DMA waits that are too loose
only show under such load."""
import pytest
import torch

from morphablediffusion_amd import synthetic
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
from tests import golden_inputs as gi
from tests.test_gpu_model import make_model, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,bvn", [(4, 4), (8, 2)])
def test_step_is_bit_reproducible(N, bvn):
    m = make_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, workspace_gb=10.0)
    batch = to_dev(synthetic.make_batch(N, "perspective", 600, mesh_seed=1))
    x_T, x_in, clip = [t.cuda() for t in synthetic.make_latents(N, 32, seed=6033)]
    noise = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(1)).cuda()

    def step():
        ts = torch.full((1,), int(m.sampler.ddim_timesteps[20]), dtype=torch.long, device="cuda")
        return m.sampler.denoise_apply(x_T, {"x": x_in}, clip, ts, 20, 2.0, batch_view_num=bvn, batch=batch, noise=noise)

    ref = step()
    bad = sum(0 if torch.equal(step(), ref) else 1 for _ in range(60))
    assert bad == 0, f"{bad} of 60 repetitions differ"
    m.engine.close()


def test_kernels_repeat_under_memory_load():
    from morphablediffusion_amd.engine import Engine
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=6.0)
    g = torch.Generator().manual_seed(0)
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(1 << 27, device="cuda"), torch.empty(1 << 27, device="cuda")
    idx = torch.randint(0, 1 << 21, (1 << 21,), device="cuda")
    tab = torch.empty(1 << 21, 64, device="cuda")

    def load():
        with torch.cuda.stream(side):
            for _ in range(4):
                tab.index_select(0, idx)
                big_a.copy_(big_b)

    x = torch.randn(8, 320, 32, 32, generator=g).cuda()
    w = (torch.randn(320, 320, 3, 3, generator=g) * 0.02).cuda()
    a = torch.randn(8192, 640, generator=g).cuda()
    wl = (torch.randn(1280, 640, generator=g) * 0.03).cuda()
    a3 = torch.randn(2048, 1280, generator=g).cuda()
    wl3 = (torch.randn(1280, 1280, generator=g) * 0.03).cuda()
    x3 = torch.randn(2, 64, 24, 32, 32, generator=g).cuda()
    w3 = (torch.randn(128, 64, 3, 3, 3, generator=g) * 0.02).cuda()
    q = torch.randn(8, 256, 128, generator=g).cuda()
    cases = {"halo conv": lambda: e.op_conv(x, w), "dma gemm": lambda: e.op_linear(a, wl, a_half=True),
             "dma gemm split-K": lambda: e.op_linear(a3, wl3, a_half=True), "conv3d s2": lambda: e.op_conv3d(x3, w3, stride=2),
             "attention": lambda: e.op_attention(q, q * 0.5, q * 0.25, 8)}
    for name, fn in cases.items():
        ref = fn()
        torch.cuda.synchronize()
        for _ in range(20):
            load()
            assert torch.equal(fn(), ref), name
        torch.cuda.synchronize()
    e.close()


def test_frustum_volumes_unchanged_beside_halo_convs_of_another_stream():
    """The isolated form of the side-stream hazard (tools/race_micro.py, tools/race_probe.hip): engine B computes the frustum
    volumes on its own stream while engine A -- another context, workspace and stream -- runs LDS-halo convolutions.  Before
    the fix 15-20 of 20 runs differed from the idle-GPU result: the gather's camera arithmetic was SLP-packed into a
    v_pk_mul_f32 with crossed half selects, whose low product comes back as 0 in lanes 48-63 when the wave shares a SIMD
    with conv3_dma_kernel waves.  Built without that instruction the volumes are bit-identical."""
    from morphablediffusion_amd.engine import Engine
    N = 16
    m = make_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, workspace_gb=12.0)
    batch = to_dev(synthetic.make_batch(N, "perspective", 5023, mesh_seed=1))
    x_T, _, _ = [t.cuda() for t in synthetic.make_latents(N, 32, seed=6033)]
    ts = torch.full((1,), 481, dtype=torch.long, device="cuda")
    t_embed, v_embed = m.embed_time(ts), m.get_viewpoint_embedding(batch)
    sv = m.spatial_volume.construct_spatial_volume(x_T, t_embed, v_embed, batch)
    idx = torch.arange(N)[None]
    frustum = lambda: m.spatial_volume.construct_view_frustum_volume(sv, t_embed, v_embed, idx, batch)[0]
    ref = frustum()
    torch.cuda.synchronize()
    eA = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(32, 320, 16, 16, generator=g).cuda()
    w = (torch.randn(1280, 320, 3, 3, generator=g) * 0.02).cuda()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for _ in range(20):
        torch.cuda.synchronize()
        with torch.cuda.stream(sA):
            for _ in range(12):
                eA.op_conv(x, w)
        with torch.cuda.stream(sB):
            out = frustum()
        torch.cuda.synchronize()
        bad += 0 if all(torch.equal(out[k], ref[k]) for k in ref) else 1
    assert bad == 0, f"{bad} of 20 frustum volumes differ from the idle-GPU result"
    eA.close()
    m.engine.close()


@pytest.mark.parametrize("env", [{"MVD_ONE_WAY_FORK": "1"}, {}], ids=["one-way-fork", "two-way-fork"])
def test_full_width_step_is_bit_reproducible_under_both_fork_orders(env):
    """Full width, N = 16: with a one-way fork the side stream's gather is dispatched onto CUs the trunk's halo convs already
    occupy -- 60-100 % of repetitions differed before the fix (profiles/r02_a_side_stream_race.txt).  The library reads the
    switch once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, DET_FULL="1", DET_N="16", DET_WS="30", DET_REPS="40", **env)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "det_step.py")], cwd=root, env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mismatches 0 of 40" in r.stdout, r.stdout[-2000:]
    print("property [determinism]", env or "default", r.stdout.strip().splitlines()[-1][:80])
