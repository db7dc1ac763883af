#!/usr/bin/env python
"""Headline benchmark: multi-view denoising steps/sec (BASELINE.json).

One "step" = one SyncDDIMSampler.denoise_apply (reference morphable_diffusion.py:701-739) over all N=16
views of one sample with classifier-free guidance (UNet batch 2N), full-width UNet, 5023-vertex FLAME-sized
synthetic mesh, 256x256 images (32x32 latents).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (views sharded)

Prints ONE JSON line (rank 0).  With N GPUs the 16 views are partitioned over the ranks (strong scaling:
total work fixed) and the only per-step exchange is one RCCL all-reduce of the [Nv,16] fused vertex features.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VIEWS = 16
PEAK_F16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0     # HBM3E, same guide


PMC_TRAFFIC_FILE = os.environ.get("MVD_PMC_TRAFFIC", os.path.join("profiles", "pmc_traffic.json"))


def pmc_traffic(family, config):
    """HBM-side bytes per launch of ``family`` from the PMC summary (tools/pmc_traffic.py ... out.json), or None."""
    try:
        with open(os.path.join(ROOT, PMC_TRAFFIC_FILE) if not os.path.isabs(PMC_TRAFFIC_FILE) else PMC_TRAFFIC_FILE) as f:
            d = json.load(f)
        if d.get("config") != config:
            return None
        from morphablediffusion_amd.lib import csrc_sha16
        if d.get("csrc_sha16") != csrc_sha16():  # counters of another build: not this run's traffic
            return None
        return d["bytes_per_launch"].get(family)
    except (OSError, ValueError, KeyError):
        return None


KERNEL_DURATIONS_FILE = os.environ.get("MVD_KERNEL_DURATIONS", os.path.join("profiles", "kernel_durations.json"))


def rocprof_us_per_launch(family, config):
    """rocprofv3's average duration (us) of ``family`` from the stamped summary tools/prof_summary.py writes (same build, same
    workload), or None."""
    try:
        with open(os.path.join(ROOT, KERNEL_DURATIONS_FILE) if not os.path.isabs(KERNEL_DURATIONS_FILE) else KERNEL_DURATIONS_FILE) as f:
            d = json.load(f)
        if d.get("config") != config:
            return None
        from morphablediffusion_amd.lib import csrc_sha16
        if d.get("csrc_sha16") != csrc_sha16():
            return None
        return d["us_per_launch"].get(family)
    except (OSError, ValueError, KeyError):
        return None


def unet_kwargs(cfg):
    return dict(volume_dims=list(cfg.volume_dims), image_size=cfg.image_size, in_channels=8, out_channels=4,
                model_channels=cfg.model_channels, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
                context_dim=768, use_checkpoint=True, legacy=False)


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pick_cpu_threads():
    """Eager PyTorch on hundreds of small ops stops scaling -- and on a many-socket host thrashes by an order of magnitude --
    well before os.cpu_count() threads.  A reduced-width UNet forward (the same op mix, ~1 s) is timed at 16, 32, 64, ...
    up to os.cpu_count() threads; the fastest count is the one the baseline runs with (reported as ``cores``)."""
    from morphablediffusion_amd.spec import UNetConfig, build_unet_plan, unet_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import mvd_oracle as O
    forced = os.environ.get("MVD_CPU_THREADS")
    if forced:
        return int(forced), {}
    ncpu = os.cpu_count() or 1
    cfg = UNetConfig(model_channels=64)
    W = seeded_state_dict(unet_manifest(cfg), 7)
    plan = build_unet_plan(cfg)
    g = torch.Generator().manual_seed(0)
    x, t, ctx = torch.randn(8, 8, 32, 32, generator=g), torch.full((8,), 481), torch.randn(8, 1, 768, generator=g)
    sd = {32 >> l: torch.randn(8, c, 48 >> l, 32 >> l, 32 >> l, generator=g) for l, c in enumerate(cfg.volume_dims)}
    cands, n = [], 16
    while n < ncpu:
        cands.append(n)
        n *= 2
    cands.append(ncpu)
    seen, best_n, best_t = {}, cands[0], None
    for n in cands:
        torch.set_num_threads(n)
        with torch.no_grad():
            O.unet_forward(W, plan, x[:2], t[:2], ctx[:2], {k: v[:2] for k, v in sd.items()})  # warm the thread pool
            t0 = time.time()
            O.unet_forward(W, plan, x, t, ctx, sd)
            dt = time.time() - t0
        seen[n] = round(dt, 3)
        if best_t is None or dt < best_t:
            best_n, best_t = n, dt
        elif dt > 1.5 * best_t and n != ncpu:
            # past the knee: more threads only thrash -- skip the points in between, but still time os.cpu_count() threads, so the
            # line carries the all-cores figure next to the best one.  That point runs in a process of its
            # own with a hard 60 s limit: on a 256-thread host it took 300+ s of this leg's 420 s budget in round 5 and would
            # have cost the whole cpu_baseline object on a slightly slower box
            import subprocess
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-thread-point", str(ncpu)], capture_output=True,
                                   text=True, timeout=60, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
                val = [l for l in p.stdout.splitlines() if l.startswith("POINT ")]
                seen[ncpu] = round(float(val[-1].split()[1]), 3) if val else None
            except subprocess.TimeoutExpired:
                seen[ncpu] = ">240 (a quarter of the batch did not finish in 60 s)"
            break
    return best_n, seen


def cpu_thread_point(n):
    """``bench.py --cpu-thread-point N``: the reduced-width UNet forward of pick_cpu_threads on a quarter of its batch with N
    threads, scaled to the whole batch -- one line ``POINT seconds``."""
    from morphablediffusion_amd.spec import UNetConfig, build_unet_plan, unet_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import mvd_oracle as O
    cfg = UNetConfig(model_channels=64)
    W = seeded_state_dict(unet_manifest(cfg), 7)
    plan = build_unet_plan(cfg)
    g = torch.Generator().manual_seed(0)
    x, t, ctx = torch.randn(2, 8, 32, 32, generator=g), torch.full((2,), 481), torch.randn(2, 1, 768, generator=g)
    sd = {32 >> l: torch.randn(2, c, 48 >> l, 32 >> l, 32 >> l, generator=g) for l, c in enumerate(cfg.volume_dims)}
    torch.set_num_threads(n)
    with torch.no_grad():
        t0 = time.time()
        O.unet_forward(W, plan, x, t, ctx, sd)
    print(f"POINT {(time.time() - t0) * 4.0:.3f}")


def cpu_baseline_main():
    """``bench.py --cpu-baseline-only``: prints the cpu_baseline object (run as a subprocess of the bench so that a stalled
    CPU run can never take the GPU result with it).  The oracle (CPU restatement, pinned to the reference's goldens: kind
    "port") on this host's cores (SURVEY 8(d) / BASELINE.md section 3): ONE full denoise_apply of the headline
    configuration (N=16 views, CFG 2.0, full-width UNet, 5023-vertex mesh, fp32 eager) and one step of the plumbing
    configuration (configs[0]: one view, 64x64 latent, first DDIM step)."""
    import dataclasses
    from morphablediffusion_amd import synthetic
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, build_unet_plan, full_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import mvd_oracle as O
    threads, calib = pick_cpu_threads()
    torch.set_num_threads(max(1, threads))
    tab = O.ddim_tables(50, 1.0)
    ucfg = UNetConfig()

    def one(N, size, index, ucfg_):
        vcfg = VolumeConfig(num_views=N, input_image_size=size)
        W_ = seeded_state_dict(full_manifest(ucfg_, vcfg), 7)
        batch = synthetic.make_batch(N, "perspective", 5023, mesh_seed=1, image_size=size)
        x_T, x_in, clip = synthetic.make_latents(N, size // 8, seed=6033)
        ts = torch.full((1,), int(tab["timesteps"][index]), dtype=torch.long)
        noise = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(1)) if index else None
        plan = build_unet_plan(ucfg_)
        t0 = time.time()
        with torch.no_grad():
            out = O.denoise_apply(W_, plan, vcfg, tab, x_T, x_in, clip, ts, index, 2.0, batch, batch_view_num=N, noise=noise)
        assert torch.isfinite(out).all()
        return time.time() - t0

    dt16 = one(N_VIEWS, 256, 49, ucfg)
    dt1 = one(1, 512, 0, dataclasses.replace(ucfg, image_size=64))
    print(json.dumps({
        "value": 1.0 / dt16, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
        "cpu_model": cpu_model_name(), "os_cpu_count": os.cpu_count(), "thread_calibration_s": calib,
        "all_threads": {"threads": os.cpu_count(), "reduced_width_unet_s": calib.get(os.cpu_count()),
                        "best_threads_reduced_width_unet_s": min(v for v in calib.values() if isinstance(v, (int, float))) if calib else None,
                        "note": "the headline-width step is only run at the best thread count: with every hardware thread "
                                "eager PyTorch is slower by the ratio of these two figures (and did not finish in 20 minutes)"},
        "sample": f"ONE full denoise_apply of the headline configuration (N={N_VIEWS}, CFG 2.0, full-width UNet, "
                  f"5023-vertex mesh, fp32 eager): {dt16:.1f} s; plus configs[0] (one view, 64x64 latent, first DDIM "
                  f"step, full width): {dt1:.1f} s.  Thread count = the fastest of 16, 32, ... os.cpu_count() on a "
                  f"reduced-width UNet forward (eager PyTorch thrashes beyond it)",
        "config0_step_s": dt1}))


def cpu_baseline(timeout_s=420):
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                           timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and line:
            return json.loads(line[-1])
        return {"value": None, "unit": "steps/s", "cores": None, "kind": "port", "sample": f"failed: {p.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "steps/s", "cores": None, "kind": "port",
                "sample": f"the CPU oracle did not finish one step within {timeout_s} s on this host"}


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks of ONE node ourselves, exactly as the driver does
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1).  Rank 0 of the child prints the JSON line on our
    stdout.  Fails loudly -- a JSON error line and a non-zero exit -- when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "MVD_FORCE_DEVICE" not in os.environ:
        print(json.dumps({"error": f"--gpus {n} requested but this node has {have} visible GPU(s)"}))
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def train_main(args):
    """``--config train`` (NOT the headline metric): BASELINE.json configs[3]'s unit of work -- one training step of
    SyncMultiviewDiffusion (training_step morphable_diffusion.py:520-549 on B samples per GPU, N = 16 views, finetune_unet) =
    conditioner + UNet forward, MSE, backward through every UNet block, ONE all-reduce of the flat gradient arena (N > 1),
    fused AdamW on the arenas, in-place re-pack of the fp16 weights; the conditioner's backward (spatial_volume / time_embed
    gradients) runs per sample.  ``prepare`` (VAE / CLIP on images) is replaced by seeded latents.  fp16 MFMA operands, fp32
    accumulation and master weights, dynamic loss scale."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("MVD_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if args.gpus != world:
        os.write(real_stdout, (json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE is {world}"}) + "\n").encode())
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("MVD_DIST_BACKEND", "nccl"))
    dev = f"cuda:{local}"
    from morphablediffusion_amd import synthetic
    from morphablediffusion_amd.model import SyncMultiviewDiffusion
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    N, B = N_VIEWS, args.train_batch
    ucfg, vcfg = UNetConfig(), VolumeConfig(num_views=N)
    W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)
    sched_cfg = {"target": "ldm.lr_scheduler.LambdaLinearScheduler",
                 "params": dict(warm_up_steps=[100], cycle_lengths=[100000], f_start=[0.02], f_max=[1.0], f_min=[1.0])}
    model = SyncMultiviewDiffusion(
        unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": unet_kwargs(ucfg)},
        scheduler_config=sched_cfg, finetune_unet=True, view_num=N, image_size=256, cfg_scale=2.0, device=dev, workspace_gb=96.0,
        train_mode=True, recompute=not args.keep_activations, precision_level=2,  # (the training tests' level; inference default: 3)
        loss_scale=1.0 if args.dtype == "bf16" else 65536.0)  # bf16 has fp32's exponent range: no loss scaling
    model.load_state_dict(W)
    from morphablediffusion_amd import lib as mvd_lib
    lib_dtype = mvd_lib.load().mvd_compute_dtype().decode()
    assert lib_dtype == args.dtype, (lib_dtype, args.dtype)
    (opt,), (sched,) = model.configure_optimizers()
    # Every step sees a NEW batch, as a training loop does: per sample another mesh (a pool of 2B synthetic meshes, cycled) and
    # another camera order, in fresh device tensors -- so the per-sample tables (sparse-conv rule book, cameras) are rebuilt
    # inside the timed step for every sample; nothing of the conditioner is carried over from the previous step.
    NV = args.train_vertices
    pool = [synthetic.make_batch(N, "perspective", NV, mesh_seed=1 + i, radii=(0.22 + 0.004 * (i % 5), 0.28, 0.25 - 0.003 * (i % 3)))
            for i in range(2 * B)]
    nv_min = min(p["vertices"].shape[1] for p in pool)  # the synthetic meshes are voxel-de-duplicated: equalise the vertex count
    for p in pool:                                      # so that samples stack (the bounding box / out_sh stay valid for a subset)
        p["vertices"], p["coord"] = p["vertices"][:, :nv_min].contiguous(), p["coord"][:, :nv_min].contiguous()

    def make_step_batch(step):
        out = {}
        for k in pool[0]:
            out[k] = torch.cat([pool[(step * B + bi) % len(pool)][k] for bi in range(B)], 0).clone()
        for bi in range(B):
            out["target_K"][bi] = out["target_K"][bi].roll(bi + rank + step, 0)
            out["target_RT"][bi] = out["target_RT"][bi].roll(bi + rank + step, 0)
        return {k: v.to(dev) for k, v in out.items()}

    step_batches = [make_step_batch(i) for i in range(args.warmup + args.steps)]
    g = torch.Generator().manual_seed(77 + rank)
    prepared = ((torch.randn(B, N, 4, 32, 32, generator=g) * 0.8).to(dev), torch.randn(B, 1, 768, generator=g).to(dev),
                {"x": (torch.randn(B, 4, 32, 32, generator=g) * 0.18215).to(dev)})
    losses = []

    def one_step():
        opt.zero_grad()
        loss = model.training_step(step_batches[len(losses)], prepared=prepared)
        model.sync_gradients()
        opt.step()
        sched["scheduler"].step()
        losses.append(loss)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lv = [float(l) for l in losses]
    assert all(v == v and abs(v) < 1e6 for v in lv), lv
    if rank == 0:
        # algorithmic work of one step: UNet forward 202.0 GFLOP per sample, backward = dgrad + wgrad = 2x,
        # + one re-run of the forward when blocks are recomputed; the conditioner's forward is not counted
        per_sample = 202.0e9 * (3 + (0 if args.keep_activations else 1))
        out = {"metric": "NOT the headline metric: training samples/sec of BASELINE config 'train' (configs[3]: N=16 views, "
                         "finetune_unet training step, full-width UNet)",
               "value": B * world * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": lib_dtype, "dtype_detail": f"{lib_dtype} MFMA operands and activation / weight storage, f32 accumulation, f32 master "
                                                   f"weights and Adam moments" + ("" if lib_dtype == "bf16" else ", dynamic loss scale"),
               "data": "synthetic",
               "config": {"workload": f"training step, {B} samples per GPU x {N} views (seeded latents instead of VAE/CLIP on "
                                      f"images; a NEW {NV}-vertex mesh and camera order per sample and step: the sparse-conv rule book "
                                      f"and camera tables are rebuilt inside the step), full-width UNet (916.9M params, random init), "
                                      f"finetune_unet=True, AdamW lr 5e-5 / 5e-4, "
                                      f"{'all activations kept' if args.keep_activations else 'per-block activation recompute'}",
                          "name": "train", "batch_per_gpu": B, "mesh_vertices_requested": NV,
                          "mesh_vertices_after_voxel_dedup": int(nv_min), "parallelism": f"data-parallel x{world}" if world > 1 else "single GPU"},
               "unet_tflops": per_sample * B * world / (dt / args.steps) / 1e12,
               "loss_first_last": [lv[0], lv[-1]], "loss_scale": model.loss_scale, "optimizer_steps_skipped": opt.steps_skipped,
               "rccl_ranks": dist.get_world_size() if world > 1 else 1, "dist_backend": dist.get_backend() if world > 1 else None}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-thread-point", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the 50-step trajectory and the VAE decode reported next to the headline (profiling runs: the "
                         "process then executes only identical denoising steps)")
    ap.add_argument("--batch-view-num", type=int, default=0, help="views per UNet pass (0 = all local views)")
    ap.add_argument("--train-batch", type=int, default=8, help="--config train: samples per GPU and step")
    ap.add_argument("--train-vertices", type=int, default=5023,
                    help="--config train: mesh vertices per sample (5023 = FLAME; FaceScape's bilinear topology has 26317)")
    ap.add_argument("--keep-activations", action="store_true",
                    help="--config train: keep every activation of the forward pass instead of re-running each block before its "
                         "backward (the reference's use_checkpoint: True is the default)")
    ap.add_argument("--dtype", default=None, choices=["f16", "bf16"],
                    help="MFMA operand / storage type = which build of the library is loaded (MVD_DTYPE).  Default: f16 for the "
                         "denoising configurations (north_star's parity bound is stated for fp16), bf16 for --config train "
                         "(BASELINE configs[3] names bf16)")
    ap.add_argument("--config", default="headline", choices=["headline", "n8", "smplx32", "train"],
                    help="headline = BASELINE.json's metric configuration (N=16, 256^2: configs[2], the default the driver "
                         "runs); n8 = configs[1] (N=8, 256^2); smplx32 = configs[4] (SMPL-X-sized mesh, N=32 views, 512^2 -> "
                         "64^2 latents, orthographic cameras, 4 views per UNet pass); train = configs[3]'s unit of work, one training step "
                         "(forward, loss, UNet backward, gradient all-reduce, AdamW, re-pack).  Only 'headline' is the headline value")
    ap.add_argument("--probe-stride", type=int, default=4,
                    help="bracket 1 in N launches of the dominant kernel family with HIP events inside the timed region")
    ap.add_argument("--simulate-gpus", type=int, default=0,
                    help="timing aid: run ONE rank's share of a G-way view sharding on one GPU (no collective); "
                         "reported as a per-rank step time, never as the headline value")
    args = ap.parse_args()
    if args.cpu_thread_point:
        cpu_thread_point(args.cpu_thread_point)
        return
    if args.cpu_baseline_only:
        return cpu_baseline_main()
    if args.dtype is None:
        args.dtype = os.environ.get("MVD_DTYPE") or ("bf16" if args.config == "train" else "f16")
    os.environ["MVD_DTYPE"] = args.dtype  # read by morphablediffusion_amd.lib at its first import (below, in every rank)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    if args.config == "train":
        return train_main(args)

    # stdout carries exactly ONE JSON line: whatever libraries print there (gloo / RCCL connection chatter, sample()'s
    # progress lines) is sent to stderr at the file-descriptor level; the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test knobs (single-GPU boxes): run the multi-rank flow with gloo and every rank on one device
    backend = os.environ.get("MVD_DIST_BACKEND", "nccl")
    if "MVD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["MVD_FORCE_DEVICE"])
    if args.gpus != world:
        # never a silent single-rank run labelled n_gpus = 1: the launcher and --gpus must agree
        os.write(real_stdout, (json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with "
                                                    f"torch.distributed.run --nproc-per-node {args.gpus} (or plain "
                                                    f"`python bench.py --gpus {args.gpus}`, which spawns the ranks itself)"}) + "\n").encode())
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend)  # "nccl" = RCCL over xGMI
        assert dist.get_world_size() == args.gpus, f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}"
    dev = f"cuda:{local}"

    from morphablediffusion_amd import synthetic
    from morphablediffusion_amd.model import SyncMultiviewDiffusion
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
    from morphablediffusion_amd.weights import seeded_state_dict

    global N_VIEWS
    CFG = {"headline": dict(N=16, size=256, proj="perspective", nverts=5023, radii=(0.22, 0.28, 0.25), bvn=0),
           "n8": dict(N=8, size=256, proj="perspective", nverts=5023, radii=(0.22, 0.28, 0.25), bvn=0),
           "smplx32": dict(N=32, size=512, proj="orthographic", nverts=10475, radii=(0.18, 0.45, 0.12), bvn=4)}[args.config]
    N_VIEWS = CFG["N"]
    ucfg = UNetConfig(image_size=CFG["size"] // 8)
    vcfg = VolumeConfig(num_views=N_VIEWS, projection=CFG["proj"], input_image_size=CFG["size"])
    W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)  # random-init weights of the reference architecture
    from morphablediffusion_amd.spec import VaeConfig, vae_decoder_manifest
    W.update(seeded_state_dict(vae_decoder_manifest(VaeConfig()), 7))  # first-stage decoder (reported separately)
    model = SyncMultiviewDiffusion(
        unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": unet_kwargs(ucfg)},
        projection=CFG["proj"], view_num=N_VIEWS, image_size=CFG["size"], cfg_scale=2.0, device=dev, workspace_gb=48.0)
    model.load_state_dict(W)
    model.eval()  # generate_face.py:77
    sampler = model.sampler
    sampler.shard_views = world > 1
    if args.simulate_gpus and world == 1:
        sampler.simulate_world = args.simulate_gpus
    lo, hi = sampler.view_range(N_VIEWS)
    nl = hi - lo
    bvn = args.batch_view_num or min(nl, CFG["bvn"] or nl)
    lat = CFG["size"] // 8

    batch = {k: v.to(dev) for k, v in synthetic.make_batch(N_VIEWS, CFG["proj"], CFG["nverts"], mesh_seed=1,
                                                            image_size=CFG["size"], radii=CFG["radii"]).items()}
    x_T, x_in, clip = [t.to(dev) for t in synthetic.make_latents(N_VIEWS, lat, seed=6033)]
    x = x_T[:, lo:hi].contiguous()
    info = {"x": x_in}
    g = torch.Generator(device=dev).manual_seed(123)
    noise = torch.randn(1, N_VIEWS, 4, lat, lat, device=dev, generator=g)[:, lo:hi].contiguous()
    nsteps = len(sampler.ddim_timesteps)

    def one_step(i, xx):
        index = nsteps - 1 - (i % nsteps)
        step = int(sampler.ddim_timesteps[index])
        ts = sampler._time_steps(1, step, dev)  # as SyncDDIMSampler.sample does: one resident tensor per DDIM step value
        return sampler.denoise_apply(xx, info, clip, ts, index, 2.0, batch_view_num=bvn, is_step0=index == 0,
                                     batch=batch, noise=noise, host_steps=[step])

    with torch.no_grad():
        for i in range(args.warmup):
            x = one_step(i, x)
        # survey pass (NOT timed): HIP events on the launch stream around EVERY launch of every kernel family for two
        # steps -> per-family table (time, algorithmic FLOPs / bytes); the family with the largest summed time is the
        # dominant kernel, and only that one is bracketed (a 1-in-stride sample of its launches) in the timed region
        model.engine.probe_config(1)
        for i in range(2):
            x = one_step(args.warmup + i, x)
        torch.cuda.synchronize()
        families = model.engine.probe_report()
        model.engine.probe_config(0)
        # what an event bracket adds to a launch's own duration: pairs of events with nothing between them, recorded by the
        # survey pass on the same stream after every 8th launch; subtracted per bracketed launch below (rocprofv3's kernel
        # durations, which the roofline must agree with, do not contain it)
        # ... measured two ways: an EMPTY bracket (event overhead alone) and a bracket around a NULL kernel (event overhead +
        # the dispatch latency between the first event and the kernel's first wave).  The second is what separates an event
        # bracket from rocprofv3's begin-to-end kernel duration (round 5 subtracted the first: 16 % high on a 45 us kernel)
        empty = [f for f in families if f["family"] == "(empty bracket)"]
        nullk = [f for f in families if f["family"] == "(null-kernel bracket)"]
        families = [f for f in families if not f["family"].startswith("(")]
        empty_ms = (empty[0]["ms"] / empty[0]["sampled"]) if empty and empty[0]["sampled"] else 0.0
        null_ms = (nullk[0]["ms"] / nullk[0]["sampled"]) if nullk and nullk[0]["sampled"] else 0.0
        bracket_ms = null_ms if null_ms > 0 else empty_ms
        for f in families:
            f["ms_raw"] = f["ms"]
            f["ms"] = max(f["ms"] - f["sampled"] * bracket_ms, 0.5 * f["ms"])
        dominant = max(families, key=lambda f: f["ms"])["family"] if families else None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if dominant:
            model.engine.probe_config(2, dominant, args.probe_stride)
        t0 = time.perf_counter()
        for i in range(args.steps):
            x = one_step(args.warmup + 2 + i, x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(x).all()

    timed = {f["family"]: f for f in model.engine.probe_report()} if dominant else {}
    model.engine.probe_config(0)
    for f in timed.values():
        f["ms_raw"] = f["ms"]
        f["ms"] = max(f["ms"] - f["sampled"] * bracket_ms, 0.5 * f["ms"])

    # reported next to the headline value ("plus 50-step wall-time"): one full 50-step DDIM trajectory
    # through SyncDDIMSampler.sample, and the first-stage decode of this rank's views
    extras = {"ddim50_wall_s": None, "vae_decode_ms": None}
    try:
        if args.no_extras:
            raise RuntimeError("skipped (--no-extras)")
        gen = torch.Generator(device=dev).manual_seed(6033)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        import contextlib
        with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):  # sample() prints like the reference does
            x50, _ = sampler.sample(info, clip, unconditional_scale=2.0, batch_view_num=bvn, batch=batch, generator=gen)
        torch.cuda.synchronize()
        extras["ddim50_wall_s"] = time.perf_counter() - t1
        zl = x50[0, lo:hi].contiguous()
        model.decode_first_stage(zl)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        img = model.decode_first_stage(zl)
        torch.cuda.synchronize()
        extras["vae_decode_ms"] = 1e3 * (time.perf_counter() - t1)
        assert torch.isfinite(img).all() and tuple(img.shape) == (nl, 3, CFG["size"], CFG["size"])
    except Exception as exc:  # never lose the headline line to an extra
        print(f"bench extras failed: {exc!r}", file=sys.stderr)

    # Roofline of the dominant kernel family (largest summed time in the survey pass).  For the bracketed launches of the
    # timed region:  t_mfma = sum(flops) / peak,  t_hbm = sum(algorithmic bytes) / peak;  the larger one is the roof that
    # bounds the family, `achieved` is the family's aggregate rate in that roof's unit, frac = achieved / peak.
    def roof(f):
        t = f["ms"] * 1e-3
        if t <= 0 or f["sampled"] == 0:
            return None
        tf, tb = f["flops"] / (PEAK_F16_TFLOPS * 1e12), f["bytes"] / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if tf >= tb else "hbm"
        return {"bound": bound, "achieved": (f["flops"] / t / 1e12) if bound == "mfma" else (f["bytes"] / t / 1e9),
                "peak": PEAK_F16_TFLOPS if bound == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                "tflops": f["flops"] / t / 1e12, "gbs": f["bytes"] / t / 1e9, "us_per_launch": 1e6 * t / f["sampled"],
                "us_per_launch_raw": 1e3 * f.get("ms_raw", f["ms"]) / f["sampled"],
                "launches_bracketed": f["sampled"], "launches": f["launches"]}

    fam_rows = []
    tot_ms = sum(f["ms"] for f in families) or 1.0
    for f in sorted(families, key=lambda f: -f["ms"]):
        r = roof(f) or {}
        fam_rows.append({"family": f["family"], "launches_per_step": f["launches"] / 2.0, "ms_per_step": f["ms"] / 2.0,
                         "share": f["ms"] / tot_ms, "tflops": r.get("tflops"), "gbs": r.get("gbs"), "bound": r.get("bound"),
                         "frac": (r["achieved"] / r["peak"]) if r else None})
    dom = roof(timed[dominant]) if dominant and dominant in timed else None
    if dom is None and families:  # nothing bracketed in the timed region (stride too large for the step count)
        dom = roof(max(families, key=lambda f: f["ms"]))
    us_rp = rocprof_us_per_launch(dominant, args.config) if (dominant and world == 1 and not args.simulate_gpus) else None
    if dom is not None:  # algorithmic work per launch in units of the bounding roof's peak-seconds: frac = this / launch duration
        dom["frac_alg_per_launch"] = dom["achieved"] / dom["peak"] * dom["us_per_launch"] * 1e-6
    if rank == 0:
        out = {
            "metric": "multi-view denoising steps/sec (N=16 views, 256x256, CFG 2.0, DDIM-50 step)" if args.config == "headline"
                      else f"NOT the headline metric: denoising steps/sec of BASELINE config '{args.config}' "
                           f"(N={N_VIEWS} views, {CFG['size']}x{CFG['size']}, CFG 2.0)",
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"synthetic sample ({'FaceScape-FLAME' if CFG['nverts'] == 5023 else 'SMPL-X'}-sized mesh, "
                                   f"{CFG['nverts']} vertices before voxel de-duplication): N={N_VIEWS} target views, "
                                   f"{CFG['size']}x{CFG['size']} (latent {lat}x{lat}), {CFG['proj']} cameras, full-width UNet "
                                   f"(916.9M params, random init), CFG 2.0, one denoise_apply per step",
                       "name": args.config, "views_per_gpu": nl, "batch_view_num": bvn,
                       "parallelism": f"view-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": None if dom is None else {
                "bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                "frac": dom["achieved"] / dom["peak"],
                # HBM-side bytes per launch of this family: read from the PMC summary that tools/pmc_traffic.py writes from separate
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this build (FETCH x 2 per the gfx950 correction); null when
                # no summary for this workload is present -- the number is never hard-coded here
                "traffic": pmc_traffic(dominant, args.config) if (world == 1 and not args.simulate_gpus) else None,
                "traffic_unit": "bytes per launch, family average, from " + PMC_TRAFFIC_FILE +
                                " (null unless that file's csrc_sha16 stamp equals this tree's library sources)",
                "kernel": dominant,
                "how": f"HIP events on the launch stream around a deterministic 1-in-{args.probe_stride} sample of this "
                       f"family's launches INSIDE the timed region ({dom['launches_bracketed']} of {dom['launches']} launches, "
                       f"{dom['us_per_launch_raw']:.1f} us per bracket, {dom['us_per_launch']:.1f} us after subtracting the "
                       f"{1e3 * bracket_ms:.1f} us a bracket around a NULL kernel measures on the same stream in the survey pass: event "
                       f"overhead + dispatch latency; an empty bracket alone measures {1e3 * empty_ms:.1f} us); achieved = "
                       f"summed algorithmic {'FLOPs' if dom['bound'] == 'mfma' else 'bytes'} / summed (event time - null-kernel bracket); "
                       f"the family was picked as the one with the largest summed time in a 2-step survey pass that brackets "
                       f"every launch of every family",
                "event_bracket_overhead_us": 1e3 * bracket_ms, "empty_bracket_us": 1e3 * empty_ms, "null_kernel_bracket_us": 1e3 * null_ms,
                "us_per_launch": dom["us_per_launch"],
                # the same family under rocprofv3 (profiles/kernel_durations.json, written by tools/prof_summary.py for THIS build
                # and workload; null when absent or stale): its average kernel duration and the roofline fraction that follows
                "us_per_launch_rocprof": us_rp,
                "frac_rocprof": (dom["frac_alg_per_launch"] / (us_rp * 1e-6)) if us_rp else None,
                "us_per_launch_uncorrected": dom["us_per_launch_raw"],
                "tflops": dom["tflops"], "gbs": dom["gbs"],
                "mfma_frac": dom["tflops"] / PEAK_F16_TFLOPS, "hbm_frac": dom["gbs"] / PEAK_HBM_GBS},
            "families": fam_rows,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "dist_backend": dist.get_backend() if world > 1 else None,
            # algorithmic FLOPs of one step as the engine books them launch by launch (survey pass, all families), not a
            # constant: the SURVEY figure 433.9 GFLOP x N holds for the 32 x 32 latents of the headline only
            "step_gflop": sum(f["flops"] for f in families) / 2.0 / 1e9,
            "step_tflops": sum(f["flops"] for f in families) / 2.0 / (dt / args.steps) / 1e12,
            "ddim50_wall_s": extras["ddim50_wall_s"], "vae_decode_ms_local_views": extras["vae_decode_ms"],
        }
        if args.simulate_gpus:
            out["metric"] = f"SIMULATED per-rank step rate of a {args.simulate_gpus}-way view sharding (one rank, no collective)"
            out["n_gpus"] = 1
        if not args.no_cpu_baseline and world == 1 and not args.simulate_gpus:
            print("[bench] GPU part done: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline")}), file=sys.stderr)
            sys.stderr.flush()
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
