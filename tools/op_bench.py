"""Operator micro-benchmarks on one MI355X, one tool:  python tools/op_bench.py {linear|linear-cold|conv3|conv3-sweep|gn|attn|vae|clip} [B]

linear       dense GEMM time vs K and epilogue at the denoiser's layer shapes (warm: back-to-back launches)
linear-cold  the same layers warm vs cold (caches evicted before every launch) -- what a layer sees in situ
conv3        3x3 halo-conv time at the UNet's shapes;  conv3-sweep re-runs it per (MVD_HALO_BN, MVD_HALO_SK) in sub-processes
gn           GroupNorm(32)+SiLU at the UNet's shapes (HIP events, no layout conversion)
attn         self-attention at the three UNet levels (includes op_attention's layout conversion)
vae / clip   first-stage decode of B views / CLIP ViT-L/14 image embedding of B images
"""
import os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import ClipConfig, UNetConfig, VaeConfig, VolumeConfig, clip_manifest, vae_decoder_manifest
from morphablediffusion_amd.weights import seeded_state_dict

CONV_SHAPES = [(32, 640, 16, 640), (32, 1280, 8, 1280), (32, 1280, 16, 640), (32, 1920, 16, 640), (32, 2560, 8, 1280), (32, 960, 16, 640),
               (32, 320, 32, 320), (32, 640, 32, 320), (4, 320, 32, 320), (4, 640, 16, 640), (4, 1280, 8, 1280)]
LIN_WARM = [(32768, K, 320, {}) for K in (64, 128, 320, 640, 1280, 2560)] + [(8192, K, 640, {}) for K in (64, 320, 640, 1280, 2560)] + [
    (32768, 320, 320, dict(resid=True)), (32768, 320, 320, dict(out_half=True)), (8192, 640, 640, dict(resid=True)),
    (8192, 640, 640, dict(out_half=True)), (2048, 1280, 1280, dict(resid=True)), (32768, 320, 2560, dict(geglu=True, out_half=True)),
    (32768, 320, 2560, dict(out_half=True)), (8192, 640, 5120, dict(geglu=True, out_half=True)), (8192, 640, 5120, dict(out_half=True)),
    (32768, 320, 640, dict(out_half=True))]
LIN_COLD = [(32768, 320, 320, {}), (32768, 320, 320, dict(resid=True)), (32768, 320, 320, dict(resid=True, bias=True, rowbias=True)),
            (32768, 320, 320, dict(out_half=True)), (32768, 1280, 320, dict(out_half=True, resid=True, bias=True)),
            (32768, 320, 960, dict(out_half=True)), (32768, 320, 2560, dict(geglu=True, out_half=True, bias=True)),
            (8192, 640, 640, dict(resid=True, bias=True, rowbias=True)), (8192, 640, 640, dict(out_half=True)),
            (8192, 2560, 640, dict(out_half=True, resid=True, bias=True)), (8192, 640, 5120, dict(geglu=True, out_half=True, bias=True)),
            (2048, 1280, 1280, dict(resid=True, bias=True, rowbias=True)), (2048, 5120, 1280, dict(out_half=True, resid=True, bias=True)),
            (2048, 1280, 10240, dict(geglu=True, out_half=True, bias=True))]
GN_SHAPES = [(32, 320, 32), (32, 640, 32), (32, 960, 32), (32, 640, 16), (32, 1280, 16), (32, 1920, 16), (32, 1280, 8), (32, 2560, 8),
             (32, 1920, 8), (32, 1280, 4), (32, 2560, 4), (4, 320, 32), (4, 640, 16), (4, 1280, 8)]


def engine(gb):
    return Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=gb)


def wall(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "linear"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    if what == "conv3-sweep":
        print("shapes:", CONV_SHAPES)
        for bn in (128, 160):
            for sk in (1, 2, 3, 4, 5, 8):
                r = subprocess.run([sys.executable, __file__, "conv3"], env=dict(os.environ, MVD_HALO_BN=str(bn), MVD_HALO_SK=str(sk)),
                                   capture_output=True, text=True)
                print(f"bn={bn} sk={sk}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}")
        return
    if what == "conv3":
        e = engine(8.0)
        print(" ".join(f"{e.bench_conv(b, c, s, s, co, iters=10) * 1e3:6.1f}" for (b, c, s, co) in CONV_SHAPES))
    elif what in ("linear", "linear-cold"):
        e = engine(8.0)
        for (M, K, N, kw) in (LIN_WARM if what == "linear" else LIN_COLD):
            w = e.bench_linear(M, K, N, iters=20, **kw)
            if what == "linear":
                print(f"M={M:6d} K={K:5d} N={N:5d} {kw}: {w * 1e3:7.1f} us {2.0 * M * K * N / w / 1e9:7.1f} TF")
            else:
                c = e.bench_linear(M, K, N, iters=10, cold=True, **kw)
                print(f"M={M:6d} K={K:5d} N={N:5d} {kw}: warm {w * 1e3:6.1f} us, cold {c * 1e3:6.1f} us", flush=True)
    elif what == "gn":
        e = engine(4.0)
        for (b, C, hw) in GN_SHAPES:
            ms = e.bench_group_norm(b, C, hw * hw)
            print(f"B={b} C={C} {hw}x{hw}: {ms * 1e3:6.1f} us {b * C * hw * hw * 6 / 1e6 / ms / 1e3:5.2f} TB/s")
    elif what == "attn":
        e = engine(4.0)
        for (b, T, heads, d) in ((32, 1024, 8, 40), (32, 256, 8, 80), (32, 64, 8, 160)):
            q, k, v = (torch.randn(b, T, heads * d, device="cuda") for _ in range(3))
            print(f"B={b} T={T} d={d}: {wall(lambda: e.op_attention(q, k, v, heads), 10) * 1e6:.0f} us per op_attention call")
    elif what == "vae":
        e = engine(24.0)
        e.load_state_dict(seeded_state_dict(vae_decoder_manifest(VaeConfig()), 7))
        z = torch.randn(B, 4, 32, 32, device="cuda") * 4
        dt = wall(lambda: e.vae_decode(z), 5)
        print(f"vae decode B={B}: {dt * 1e3:.2f} ms  {622e9 * B / dt / 1e12:.0f} TFLOP/s")
    elif what == "clip":
        e = engine(2.0)
        e.load_state_dict(seeded_state_dict(clip_manifest(ClipConfig()), 0))
        x = (torch.rand(B, 3, 256, 256) * 2 - 1).cuda()
        ms = wall(lambda: e.clip_encode(x), 20) * 1e3
        gf = B * (2 * 256 * 588 * 1024 + 24 * (2 * 257 * 1024 * 1024 * 12 + 4 * 257 * 257 * 1024)) / 1e9
        print(f"clip_encode B={B}: {ms:.2f} ms  ({gf / ms:.1f} TFLOP/s)")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
