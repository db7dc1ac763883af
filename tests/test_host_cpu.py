"""CPU: host logic, the C-ABI library (loads and exports every symbol include/mvd.h declares -- no compute
calls without a GPU), schedules, block plan, synthetic batch schema, and the view-sharded sampler over a
2-rank gloo group with a test-only backend."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from morphablediffusion_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "mvd.h")).read()
    declared = sorted(set(re.findall(r"\b(mvd_[a-z0-9_]+)\s*\(", header)))
    assert "mvd_denoise_views" in declared and len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"libmvd_hip.so does not export {name}"
    assert sorted(lib.SYMBOLS) == declared


def test_library_has_no_crossed_packed_f32_multiply(tmp_path):
    """ISA lint of the built library.  On MI355X a ``v_pk_mul_f32`` whose low result takes the HIGH half of a VGPR operand
    (``op_sel:[0,1]``; the SLP vectoriser emits it for 3x4 camera transforms) returns 0 in the low half of lanes 48-63
    whenever the wave shares a SIMD with ``conv3_dma_kernel`` waves of another stream -- the "lost 128-byte lines" of the
    frustum gather (tools/race_probe.hip reproduces it with the single instruction).  The two
    kernels that had it are built with -fno-slp-vectorize; nothing in the library may (re)acquire the form."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(ROOT, "morphablediffusion_amd", "libmvd_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("llvm-objdump or the built library is missing")
    work = tmp_path / "lint"
    work.mkdir()
    shutil.copy(so, work / "libmvd_hip.so")
    subprocess.run([objdump, "--offloading", "libmvd_hip.so"], cwd=work, check=True, capture_output=True)
    so16 = os.path.join(ROOT, "morphablediffusion_amd", "libmvd_hip_bf16.so")  # the bfloat16 build of the same sources
    if os.path.exists(so16):
        shutil.copy(so16, work / "libmvd_hip_bf16.so")
        subprocess.run([objdump, "--offloading", "libmvd_hip_bf16.so"], cwd=work, check=True, capture_output=True)
    bundles = sorted(f for f in os.listdir(work) if f.endswith("gfx950"))
    assert bundles, "no gfx950 code objects found in libmvd_hip.so"
    bad, kernels, cur = [], 0, "?"
    for b in bundles:
        dis = subprocess.run([objdump, "-d", b], cwd=work, check=True, capture_output=True, text=True).stdout
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = m.group(1)
                kernels += 1
                continue
            if re.search(r"\bv_pk_(mul|fma|add)_f32\b", line) and re.search(r"op_sel:\[0,1", line) and \
                    not re.search(r"v_pk_\w+ v\[\d+:\d+\], s\[", line):
                bad.append((cur, line.split("//")[0].strip()))
    assert kernels > 100, "disassembly looks empty"
    assert not bad, f"crossed packed-f32 arithmetic on VGPR operands: {bad[:4]}"


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no GPU / extension (never route through the oracle)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from morphablediffusion_amd import lib
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    with pytest.raises(lib.MvdError):
        Engine(UNetConfig(model_channels=64), VolumeConfig())
    src = "".join(open(os.path.join(ROOT, "morphablediffusion_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "morphablediffusion_amd")) if f.endswith(".py"))
    assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), "the product package must not import the oracle"


def test_struct_layouts_match_header():
    from morphablediffusion_amd import lib
    assert ctypes.sizeof(lib.UNetConfigC) == 4 * 16
    assert ctypes.sizeof(lib.VolumeConfigC) == 4 * 14


def test_ddim_schedule_matches_reference_tables():
    from morphablediffusion_amd.schedule import DDIMSchedule
    g = np.load(os.path.join(ROOT, "tests", "golden", "ddim.npz"))
    s = DDIMSchedule(50, 1.0)
    assert np.array_equal(s.ddim_timesteps, g["timesteps"])
    assert np.array_equal(s.ddim_alphas.numpy(), g["alphas"])
    assert np.array_equal(s.ddim_alphas_prev.numpy(), g["alphas_prev"])
    assert np.array_equal(s.ddim_sigmas.numpy(), g["sigmas"])
    c = s.coefficients(49)
    assert abs(c[0] - float(g["sqrt_one_minus_alphas"][49])) < 1e-7 and c[4] == float(g["sigmas"][49])
    with pytest.raises(NotImplementedError):
        from morphablediffusion_amd.schedule import make_ddim_timesteps
        make_ddim_timesteps(50, 1000, "quad2")


def test_unet_plan_structure():
    from morphablediffusion_amd.spec import UNetConfig, build_unet_plan, unet_manifest
    plan = build_unet_plan(UNetConfig())
    assert len(plan.input_blocks) == 12 and len(plan.output_blocks) == 12 and len(plan.conditions) == 10
    kinds = [[o.kind for o in b] for b in plan.output_blocks]
    assert kinds[2] == ["res", "up"] and kinds[5] == ["res", "st", "up"] and kinds[11] == ["res", "st"]
    assert plan.output_blocks[5][0].cin == 1920 and plan.output_blocks[9][0].cin == 960
    n = sum(int(np.prod(s)) for s in unet_manifest(UNetConfig()).values())
    assert abs(n / 1e6 - 916.85) < 0.05
    with pytest.raises(NotImplementedError):
        UNetConfig(use_spatial_transformer=False).validate()


def test_batch_schema_and_voxelisation():
    from morphablediffusion_amd import synthetic
    b = synthetic.make_batch(16, "perspective", 5023, mesh_seed=1)
    Nv = b["vertices"].shape[1]
    assert b["target_K"].shape == (1, 16, 4, 4) and b["target_RT"].shape == (1, 16, 3, 4)
    assert b["coord"].shape == (1, Nv, 3) and b["coord"].dtype == torch.int32 and b["out_sh"].dtype == torch.int32
    assert (b["out_sh"] % 4 == 0).all() and (b["coord"] >= 0).all() and (b["coord"] < b["out_sh"][:, None]).all()
    key = (b["coord"][0, :, 0].long() * 4096 + b["coord"][0, :, 1]) * 4096 + b["coord"][0, :, 2]
    assert key.unique().numel() == Nv  # de-duplicated voxels
    assert b["vertices"].abs().max() < 0.5
    # camera arc: every camera looks at the origin from radius 4.5
    RT = b["target_RT"][0]
    pos = -(RT[:, :, :3].transpose(1, 2) @ RT[:, :, 3:])[:, :, 0]
    assert torch.allclose(pos.norm(dim=1), torch.full((16,), 4.5), atol=1e-4)


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from morphablediffusion_amd.model import SyncDDIMSampler
from morphablediffusion_amd import synthetic

class FakeEngine:
    """Test-only stand-in for the HIP engine with the same call surface; arithmetic is trivial but
    view-order sensitive, so a wrong partition / reduction / index mapping changes the result."""
    def __init__(self, N): self.N = N; self.num_vertices = 7
    def vertex_features(self, x, t_embed, v_embed, view_idx, add_bias=True):
        w = (view_idx.float() + 1.0).view(-1, 1)
        f = (x.reshape(x.shape[0], -1)[:, :16] * w).sum(0, keepdim=True).repeat(7, 1) / self.N
        return f + (0.5 if add_bias else 0.0)
    def vertex_view_features(self, x, t_embed, v_embed, view_idx, out=None):
        w = (view_idx.float() + 1.0).view(-1, 1, 1)
        return (x.reshape(x.shape[0], 1, -1)[:, :, :16] * w).repeat(1, 7, 1)
    def fuse_vertex_features(self, vf_all, out=None):
        acc = torch.zeros_like(vf_all[0])
        for v in range(vf_all.shape[0]): acc = acc + vf_all[v]  # fixed view order
        return acc / self.N + 0.5
    def volume_from_fused(self, fused, want_output=True, train=False): self.fused = fused.clone()
    def denoise_views(self, x, x_input, clip, timestep, t_embed, v_embed, view_idx, cfg, noise, coef, want_eps=False, out=None,
                      eps_out=None):  # (returns its own tensor: the sampler copies it into its `out` slice)
        s = self.fused.sum()
        out = x * coef[2] + s * 1e-3 + view_idx.float().view(-1, 1, 1, 1) * 1e-2
        return out + (0 if noise is None else coef[4] * noise)

class FakeModel:
    num_timesteps = 1000
    def __init__(self, N):
        self.view_num = N; self.engine = FakeEngine(N); self.device = torch.device("cpu")
        class SV:
            def _set_sample(self, batch, bi): pass
            def invalidate(self): pass
        self.spatial_volume = SV()
    def get_viewpoint_embedding(self, batch): return torch.zeros(1, self.view_num, 4)
    def embed_time(self, t): return torch.zeros(t.shape[0], 256)

WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NV = int(sys.argv[3]) if len(sys.argv) > 3 else 8

def run(shard, exchange="all_gather"):
    N = NV
    m = FakeModel(N)
    s = SyncDDIMSampler(m, 50, shard_views=shard, exchange=exchange)
    g = torch.Generator().manual_seed(7)
    x, _ = s.sample({"x": torch.zeros(1, 4, 32, 32)}, torch.zeros(1, 1, 768), unconditional_scale=2.0,
                    batch_view_num=2, batch=synthetic.make_batch(N, "perspective", 50), generator=g)
    return x

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=WORLD)
sharded = run(True)
single = run(False)
assert sharded.shape == single.shape == (1, NV, 4, 32, 32)
assert torch.equal(sharded, single)  # all-gather + fixed view-order sum: bit-identical to the single-rank trajectory
legacy = run(True, "all_reduce")
err = ((legacy - run(False, "all_reduce")).abs().max() / single.abs().max()).item()  # fp32 summation order differs
assert err < 1e-5, err
lo, hi = SyncDDIMSampler(FakeModel(NV), 50, shard_views=True).view_range(NV)
per = NV // WORLD
assert (lo, hi) == (dist.get_rank() * per, (dist.get_rank() + 1) * per)  # contiguous slices: rank g owns views [g N/G, (g+1) N/G)
print("RANK_OK", dist.get_rank(), err)
dist.destroy_process_group()
'''


def test_view_sharded_sampler_two_ranks_gloo(tmp_path):
    """N>1 path: 2 ranks x 4 views must reproduce the single-rank 8-view trajectory (same RNG stream, one
    all-reduce of the fused vertex features per step, final all-gather)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "port": 29731})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-2000:]


@pytest.mark.parametrize("views", [16, 32])
def test_view_sharded_sampler_eight_ranks_gloo(tmp_path, views):
    """The partitions BASELINE.json names, instantiated: 8 ranks x 2 views (configs[2], N = 16) and 8 ranks x 4 views (configs[4],
    N = 32) -- view_range, the full-size noise draw sliced per rank, the placement of each rank's slice in the all-gather, the
    fixed view-order sum and the final gather must reproduce the single-rank 50-step trajectory BIT FOR BIT (engine stand-in
    with view-order-sensitive arithmetic)."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER % {"root": ROOT, "port": 29751 + views})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "8", str(views)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")) for r in range(8)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-2000:]


def test_decode_first_stage_dispatch_and_vae_manifest():
    """decode_first_stage (morphable_diffusion.py:468-471) divides by the scale factor and uses the engine's decoder
    when first_stage_model.decoder.* was loaded, else the injected module; the decoder manifest has the reference's
    key set (checked against the imported reference by tools/make_goldens.py --only-vae)."""
    from morphablediffusion_amd.model import SyncMultiviewDiffusion
    from morphablediffusion_amd.spec import VaeConfig, vae_decoder_manifest

    class Eng:
        has_vae_decoder = True
        def vae_decode(self, z):
            self.seen = z
            return z.repeat(1, 1, 8, 8)[:, :3]

    class Inj:
        def decode(self, z):
            self.seen = z
            return z

    class Stub:
        first_stage_scale_factor = 0.18215
    s = Stub()
    s.engine, s.first_stage_model = Eng(), Inj()
    z = torch.ones(2, 4, 4, 4)
    out = SyncMultiviewDiffusion.decode_first_stage(s, z)
    assert out.shape == (2, 3, 32, 32) and torch.allclose(s.engine.seen, z / 0.18215)
    s.engine.has_vae_decoder = False
    SyncMultiviewDiffusion.decode_first_stage(s, z)
    assert torch.allclose(s.first_stage_model.seen, z / 0.18215)
    s.first_stage_model = None
    with pytest.raises(RuntimeError):
        SyncMultiviewDiffusion.decode_first_stage(s, z)
    man = vae_decoder_manifest(VaeConfig())
    assert len(man) == 140
    assert man["first_stage_model.decoder.conv_in.weight"] == (512, 4, 3, 3)
    assert man["first_stage_model.decoder.up.1.block.0.nin_shortcut.weight"] == (256, 512, 1, 1)
    assert man["first_stage_model.decoder.up.3.upsample.conv.weight"] == (512, 512, 3, 3)
    assert "first_stage_model.decoder.up.0.upsample.conv.weight" not in man
    assert sum(int(np.prod(v)) for v in man.values()) == 49_490_199  # decoder + post_quant_conv parameters


def test_depth_transformer_gradient_sensitivity():
    """Why the DepthTransformer gradients of the GPU training test are bounded at 5e-2 and not at 1e-2: in the REFERENCE
    arithmetic itself (fp32 oracle, autograd) a relative perturbation of 1e-3 of the block's input -- what the fp16-operand
    forward pass leaves in the activations -- moves the gradients upstream of the block's ReLU masks / depth softmax by an order
    of magnitude more than it moves the input, while a smooth (SiLU-only) ResBlock responds proportionally."""
    import torch
    from oracle import mvd_oracle as O
    from morphablediffusion_amd.spec import VolumeConfig
    from tests import golden_inputs as gi
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=4)
    W = gi.full_weights(ucfg, vcfg)
    P = "model.diffusion_model."
    gen = torch.Generator().manual_seed(3)

    def grads(fn, keys, x, *rest):
        Wl = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in W.items()}
        fn(Wl, x, *rest).backward(dout)
        return {k: Wl[k].grad for k in keys}

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()

    eps = 1e-3
    # DepthTransformer at 16x16 (output_conditions.4): ReLU x3 + depth softmax
    pre = P + "output_conditions.4"
    keys = [k for k in W if k.startswith(pre + ".")]
    x = torch.randn(2, 128, 16, 16, generator=gen)
    ctx = torch.randn(2, 128, 24, 16, 16, generator=gen) * 0.7
    dout = torch.randn(2, 128, 16, 16, generator=gen)
    noise = torch.randn(x.shape, generator=gen)
    f = lambda Wl, x_, c_: O.depth_transformer(Wl, pre, x_, c_)
    g0, g1 = grads(f, keys, x, ctx), grads(f, keys, x + eps * noise * x.std(), ctx)
    cond_move = {k[len(pre) + 1:]: rel(g1[k], g0[k]) for k in keys}
    # a ResBlock at the same place (SiLU only)
    pre_r = P + "output_blocks.7.0"
    keys_r = [k for k in W if k.startswith(pre_r + ".")]
    xr = torch.randn(2, 256, 16, 16, generator=gen)
    emb = torch.randn(2, 256, generator=gen)
    dout = torch.randn(2, 128, 16, 16, generator=gen)
    nr = torch.randn(xr.shape, generator=gen)
    fr = lambda Wl, x_, e_: O.res_block(Wl, pre_r, x_, e_)
    r0, r1 = grads(fr, keys_r, xr, emb), grads(fr, keys_r, xr + eps * nr * xr.std(), emb)
    res_move = {k[len(pre_r) + 1:]: rel(r1[k], r0[k]) for k in keys_r}
    up = max(cond_move[k] for k in ("proj_in.0.weight", "depth_attn.to_q.weight", "depth_attn.to_k.weight"))
    print(f"[sensitivity] input perturbation {eps:.0e}: DepthTransformer gradients move by up to {max(cond_move.values()):.2e} "
          f"(upstream of the softmax {up:.2e}); ResBlock gradients by up to {max(res_move.values()):.2e}")
    assert max(res_move.values()) <= 5 * eps            # smooth block: proportional
    assert up >= 8 * eps                                 # masked / softmax block: an order of magnitude more


GRAD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from morphablediffusion_amd.model import sync_flat_gradients, LambdaLinearScheduler

rank = int(sys.argv[1])
assert sync_flat_gradients(torch.ones(4)) is False          # no process group: a no-op
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=2)
# the flat gradient arena of each rank: parameter views alias it, so ONE collective averages every parameter's gradient
flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
views = [flat[0:10].view(2, 5), flat[64:64 + 300].view(3, 100), flat[512:1000]]
assert sync_flat_gradients(flat) is True
want = torch.arange(1000, dtype=torch.float32) * 1.5        # mean of x1 and x2
assert torch.equal(flat, want)
assert torch.equal(views[1], want[64:364].view(3, 100))      # the views saw the averaged values
# inf on one rank reaches every rank (the overflow check after the all-reduce then skips the step everywhere)
flat[7] = float("inf") if rank == 1 else 1.0
sync_flat_gradients(flat)
assert torch.isinf(flat[7])
s = LambdaLinearScheduler(warm_up_steps=[100], cycle_lengths=[100000], f_start=[0.02], f_max=[1.0], f_min=[1.0])
assert abs(s.schedule(0) - 0.02) < 1e-12 and abs(s.schedule(50) - 0.51) < 1e-12 and s.schedule(100) == 1.0 and s.schedule(5000) == 1.0
print("GRAD_OK", rank)
dist.destroy_process_group()
'''


BUCKET_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from morphablediffusion_amd.model import BucketedGradSync, sync_flat_gradients

rank = int(sys.argv[1])
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=WORLD)
g = torch.Generator().manual_seed(100 + rank)
local = torch.randn(4096, generator=g)
# buckets in completion order, ranges out of address order, a late bucket made of scattered pieces; [3000, 4096) is in no
# bucket (the conditioner's parameters: reduced by finish())
buckets = [[(1024, 512), (2048, 256)], [(0, 1024)], [(1536, 512), (2304, 696)]]
waited = []
flat = local.clone()
sync = BucketedGradSync(flat, buckets, wait=lambda k: waited.append(k))
sync.start()
assert waited == [0, 1, 2]
assert sync.uncovered() == [(3000, 1096)]
# after start(): bucket ranges hold the SUM over ranks, the rest is still local
everyone = [torch.randn(4096, generator=torch.Generator().manual_seed(100 + r)) for r in range(WORLD)]
total = torch.stack(everyone).sum(0)
if WORLD == 2:   # two addends: the sum is order-independent, so the collective's result is known bit for bit
    assert torch.equal(flat[0:3000], total[0:3000])
assert torch.allclose(flat[0:3000], total[0:3000], rtol=0, atol=1e-5) and torch.equal(flat[3000:], local[3000:])
assert sync.finish() is True and sync.finish() is True       # idempotent
ref = local.clone()
sync_flat_gradients(ref)
def same(a, b):
    # two ranks: a two-term sum has one order -> element for element the flat all-reduce's result.  More ranks: a ring / tree
    # all-reduce sums a CHUNK's terms in an order that depends on where the chunk lies in its message (gloo and RCCL alike), so
    # bucket-sized and arena-sized messages agree to fp32 rounding only
    return torch.equal(a, b) if WORLD == 2 else torch.allclose(a, b, rtol=1e-6, atol=1e-6)
assert same(flat, ref)
assert torch.allclose(flat, total / WORLD, rtol=1e-6, atol=1e-6)   # the mean over ranks: 1 / world applied exactly once
gathered = [torch.empty_like(flat) for _ in range(WORLD)]
dist.all_gather(gathered, flat)
assert all(torch.equal(gathered[0], t) for t in gathered)          # every rank ends with the same gradients, bit for bit
try:
    BucketedGradSync(flat, [[(0, 100)], [(50, 100)]]).uncovered()
    raise SystemExit("overlap not detected")
except RuntimeError:
    pass
# the model-level glue (training_step -> _start_grad_sync, sync_gradients, no_sync) on a stand-in with the engine's two members
import types
from morphablediffusion_amd.model import SyncMultiviewDiffusion as M
class Eng:
    def __init__(self, flat): self.flat_grads = flat
    def grad_buckets(self): return buckets
fake = types.SimpleNamespace(overlap_grad_sync=True, _grad_sync=None, _grad_comm=None, engine=Eng(local.clone()))
M._start_grad_sync(fake)
assert fake._grad_sync is not None
assert M.sync_gradients(fake) is True and fake._grad_sync is None      # consumed: one averaging per backward pass
assert same(fake.engine.flat_grads, ref)
fake.engine.flat_grads.copy_(local)
with M.no_sync(fake):                                                 # gradient accumulation: nothing is reduced
    M._start_grad_sync(fake)
    assert fake._grad_sync is None
    # ... and a sync_gradients() call inside the context (a loop that calls it after every micro-batch) is a no-op too,
    # as under DistributedDataParallel.no_sync(): no collective, no 1 / world scaling of the half-accumulated arena
    assert M.sync_gradients(fake) is False and torch.equal(fake.engine.flat_grads, local)
assert fake.overlap_grad_sync is True and torch.equal(fake.engine.flat_grads, local)
fake.overlap_grad_sync = False                                        # the flat form
M._start_grad_sync(fake)
assert fake._grad_sync is None and M.sync_gradients(fake) is True and same(fake.engine.flat_grads, ref)
print("BUCKET_OK", rank)
dist.destroy_process_group()
'''


def test_bucketed_gradient_allreduce_equals_flat_two_ranks_gloo(tmp_path):
    """DDP's bucketed reducer (BucketedGradSync: what training_step starts behind the backward pass's events) against the flat
    all-reduce: 2 gloo ranks, buckets in completion order with scattered ranges, an uncovered tail reduced by finish()."""
    script = tmp_path / "bworker.py"
    script.write_text(BUCKET_WORKER % {"root": ROOT, "port": 29743})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"BUCKET_OK {r}" in o, o[-2000:]


def test_bucketed_gradient_allreduce_equals_flat_eight_ranks_gloo(tmp_path):
    """The same reducer at the world size of BASELINE configs[3] (8 ranks): bucketed == flat element for element on every rank, the
    uncovered tail reduced by finish(), no_sync() suppresses every collective, 1 / 8 applied once."""
    script = tmp_path / "bworker8.py"
    script.write_text(BUCKET_WORKER % {"root": ROOT, "port": 29747})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")) for r in range(8)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"BUCKET_OK {r}" in o, o[-2000:]


def test_flat_gradient_allreduce_two_ranks_gloo(tmp_path):
    """DDP's gradient averaging (train_morphable_diffusion.py:302-303) as ONE all-reduce over the flat gradient arena:
    2 gloo ranks, parameter-shaped views of the buffer see the mean; the LR schedule of configs/facescape.yaml:17-24."""
    script = tmp_path / "gworker.py"
    script.write_text(GRAD_WORKER % {"root": ROOT, "port": 29741})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"GRAD_OK {r}" in o, o[-2000:]


def _rulebook_reference(coord, out_sh):
    """Independent restatement of the sparse-conv rule book (SubMConv3d k3 / SparseConv3d k3 s2 p1, three levels): dicts and
    sorted sets, the way engine_cond.hip's first version did it."""
    sites = [tuple(int(v) for v in r) for r in coord.tolist()]
    shape = [int(v) for v in out_sh.tolist()]
    subm, down, n_sites = [], [], []
    for lvl in range(3):
        idx = {}
        for i, s in enumerate(sites):
            idx.setdefault(s, i)  # first row of a voxel is its representative
        n_sites.append(len(sites))
        subm.append([[idx.get((s[0] + dz, s[1] + dy, s[2] + dx), -1) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
                     for s in sites])
        if lvl == 2:
            grid = -np.ones(shape, dtype=np.int32)
            for i, s in enumerate(sites):
                grid[s] = i
            break
        oshape = [(v - 1) // 2 + 1 for v in shape]
        outs = set()
        for s in sites:
            for k in range(27):
                n = (s[0] + 1 - k // 9, s[1] + 1 - (k // 3) % 3, s[2] + 1 - k % 3)
                if any(v % 2 or v < 0 for v in n):
                    continue
                o = tuple(v // 2 for v in n)
                if all(o[a] < oshape[a] for a in range(3)):
                    outs.add(o)
        osites = sorted(outs)
        down.append([[idx.get((2 * o[0] - 1 + k // 9, 2 * o[1] - 1 + (k // 3) % 3, 2 * o[2] - 1 + k % 3), -1) for k in range(27)]
                     for o in osites])
        sites, shape = osites, oshape
    return n_sites, subm, down, grid


@pytest.mark.parametrize("force_hash", [0, 1])
def test_rulebook_host_build_matches_reference_restatement(force_hash):
    """The host rule-book builder behind mvd_set_mesh (dense-grid path and the hash path of very large grids) against an
    independent dict-based restatement, on a mesh with duplicate voxels and sites on the grid border."""
    import ctypes as C
    from morphablediffusion_amd import lib as L, synthetic
    lib = L.load()
    verts = synthetic.ellipsoid_mesh(700, 5, (0.22, 0.28, 0.25))
    coord, out_sh, _ = synthetic.voxelize(verts)
    coord = torch.cat([coord, coord[:40]], 0).to(torch.int32).contiguous()  # duplicates: the first row must win
    out_sh = out_sh.to(torch.int32).contiguous()
    n_sites = (C.c_int32 * 3)()
    lens = (C.c_int64 * 6)()
    L.check(lib.mvd_rulebook_build(L.ptr(coord), L.ptr(out_sh), coord.shape[0], force_hash, n_sites, lens))
    ref_n, ref_subm, ref_down, ref_grid = _rulebook_reference(coord, out_sh)
    assert list(n_sites) == ref_n
    tabs = []
    for w in range(6):
        t = torch.empty(max(int(lens[w]), 1), dtype=torch.int32)
        L.check(lib.mvd_rulebook_table(w, L.ptr(t)))
        tabs.append(t[:int(lens[w])])
    for l in range(3):
        assert torch.equal(tabs[l].view(-1, 27), torch.tensor(ref_subm[l], dtype=torch.int32))
    for l in range(2):
        assert torch.equal(tabs[3 + l].view(-1, 27), torch.tensor(ref_down[l], dtype=torch.int32))
    assert torch.equal(tabs[5], torch.from_numpy(ref_grid).reshape(-1))
    bad = coord.clone()
    bad[3, 2] = out_sh[2]
    assert lib.mvd_rulebook_build(L.ptr(bad), L.ptr(out_sh), bad.shape[0], force_hash, n_sites, lens) != 0


def test_bench_reads_roofline_traffic_from_the_pmc_summary(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from the JSON tools/pmc_traffic.py writes out of the rocprofv3 --pmc passes -- for the
    same workload only, null otherwise; never a constant in the script."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    f = tmp_path / "pmc.json"
    from morphablediffusion_amd.lib import csrc_sha16
    f.write_text(json.dumps({"config": "headline", "csrc_sha16": csrc_sha16(), "bytes_per_launch": {"gemm_dma_kernel<128,0>": 1.25e8}}))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(f))
    assert bench.pmc_traffic("gemm_dma_kernel<128,0>", "headline") == 1.25e8
    stale = tmp_path / "stale.json"  # counters of another build of the library: refused
    stale.write_text(json.dumps({"config": "headline", "csrc_sha16": "0" * 16, "bytes_per_launch": {"gemm_dma_kernel<128,0>": 1.25e8}}))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(stale))
    assert bench.pmc_traffic("gemm_dma_kernel<128,0>", "headline") is None
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(f))
    assert bench.pmc_traffic("gemm_dma_kernel<128,0>", "n8") is None      # a summary of another workload
    assert bench.pmc_traffic("attn_kernel", "headline") is None           # family not in the summary
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic("gemm_dma_kernel<128,0>", "headline") is None
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert committed["config"] == "headline" and "FETCH_SIZE" in committed["source"]


def test_lds_layouts_of_the_attention_and_conv3x_kernels_are_conflict_free_in_the_bank_model():
    """The two LDS layouts round 5 changed, pinned in the bank model of tools/lds_bank_model.py (the banking
    rules of the microarchitecture guide): the attention kernel's row-major V image at the row pitches k_attn.hip selects serves
    every transposing read in one cycle per 32-lane group, conv3x's position-based XOR key every activation fragment read in one
    cycle per 16-lane group -- and the layouts they replaced do not (a change of either constant has to re-run the model)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(root, "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = open(os.path.join(root, "morphablediffusion_amd", "csrc", "k_attn.hip")).read()
    assert "constexpr int VLD = DVP <= 32 ? 32 : (DVP <= 96 ? 96 : 160);" in src, "k_attn.hip: V row pitch changed -- re-run the model"
    for D in (8, 16, 32, 40, 64, 80, 160):
        dvp = (D + 31) // 32 * 32
        vld = 32 if dvp <= 32 else (96 if dvp <= 96 else 160)
        reads = 4 * (dvp // 32) * 2            # 16-key steps x fragments x two reads
        assert m.attn_tr_read_cycles(D, vld) == 2 * reads, (D, vld)   # two 32-lane groups, one cycle each
    assert m.attn_tr_read_cycles(40, 72) > m.attn_tr_read_cycles(40, 96)
    # the transposed image of rounds 1-5: 72 -> 68 halfs halved its LDS cycles
    assert m.attn_vt_write_cycles(40, 72) + 4 * m.attn_vt_read_cycles(40, 72) == 912
    assert m.attn_vt_write_cycles(40, 68) + 4 * m.attn_vt_read_cycles(40, 68) == 496
    csrc = open(os.path.join(root, "morphablediffusion_amd", "csrc", "k_conv3x.hip")).read()
    assert "if constexpr (IW == 16) return (hx >> 1) & 7;" in csrc and "return ((hx >> 1) + 4 * (hy & 1)) & 7;" in csrc
    for iw in (16, 8):
        assert m.conv3x_halo_read_cycles(m.conv3x_key_position(iw), iw) == 1.0
        assert m.conv3x_halo_read_cycles(m.conv3x_key_linear(iw), iw) >= 2.0
