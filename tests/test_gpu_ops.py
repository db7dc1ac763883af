"""GPU: single-kernel parity through the C ABI (mvd_op_* hooks) against plain PyTorch fp32 of the same op.
Operands are rounded to fp16 inside the kernels (fp32 accumulate), so the stated tolerance is
relative-L2 <= 1e-3 (north_star: "<= 1e-3 rel fp16") and normalised max error <= 4e-3 per op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL_L2 = 1e-3
MAX_N = 4e-3


@pytest.fixture(scope="module")
def eng():
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=2.0)
    yield e
    e.close()


def close(got, want, name, rel=REL_L2, mx=MAX_N):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    rl2 = ((got - want).norm() / (want.norm() + 1e-20)).item()
    mxe = ((got - want).abs().max() / (want.abs().max() + 1e-20)).item()
    print(f"[parity] {name}: relL2={rl2:.2e} maxnorm={mxe:.2e}")
    assert rl2 <= rel and mxe <= mx, f"{name}: relL2={rl2:.3e} maxnorm={mxe:.3e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,K,N", [(300, 320, 200), (64, 1280, 1280), (1, 64, 8), (1024, 64, 384), (130, 2560, 72)])
def test_linear(eng, M, K, N):
    a, w, b = rnd(M, K), rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
    close(eng.op_linear(a, w, b), F.linear(a, w, b), f"linear {M}x{K}x{N}")


def test_linear_asymmetric_identity(eng):
    """A = I with an asymmetric B catches a transposed C write (guide: always A=I-check with asymmetric B)."""
    K = 128
    w = torch.arange(K * K, dtype=torch.float32).reshape(K, K) % 251 / 64.0
    close(eng.op_linear(torch.eye(K), w), w.t().contiguous(), "linear identity", rel=1e-6, mx=1e-6)


# dense LDS-DMA GEMM (fp16 A in HBM, M >= 512): M / N / K tails, both column widths, multi-tile walks, split-K,
# residual, GEGLU
@pytest.mark.parametrize("M,K,N,res,sk,geglu", [
    (1024, 64, 384, False, 0, False), (700, 328, 200, True, 0, False), (4096, 320, 320, True, 0, False),
    (2048, 1280, 1280, False, 0, False), (2048, 1280, 640, True, 3, False), (513, 72, 964, False, 0, False),
    (8192, 320, 2560, False, 0, True), (1000, 128, 512, False, 0, True), (16384, 64, 2048, True, 0, False),
    # weights >> activations: the launcher walks these column-group-major (xcd_prefers_cols, common.h) -- ragged M, GEGLU, split-K
    (2048, 1280, 3840, True, 0, False), (1300, 1280, 5120, False, 0, True), (512, 2560, 2560, True, 4, False),
])
def test_linear_dense(eng, M, K, N, res, sk, geglu):
    a, w, b = rnd(M, K), rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
    want = F.linear(a, w, b)
    if geglu:
        x, gate = want.chunk(2, -1)
        want = x * F.gelu(gate)
    r = rnd(*want.shape, seed=5) if res else None
    if r is not None:
        want = want + r
    got = eng.op_linear(a, w, b, geglu=geglu, resid=r, a_half=True, force_splitk=sk)
    close(got, want, f"dense linear {M}x{K}x{N} res={res} sk={sk} geglu={geglu}")


def test_linear_dense_identity(eng):
    K = 512
    w = torch.arange(K * K, dtype=torch.float32).reshape(K, K) % 251 / 64.0
    close(eng.op_linear(torch.eye(K), w, a_half=True), w.t().contiguous(), "dense linear identity", rel=1e-6, mx=1e-6)


@pytest.mark.parametrize("bn", [96, 160])
def test_linear_dense_through_the_four_wave_tiles(bn):
    """Round 6: gemm_dma_kernel<128, BN> (128-row tiles, four waves) forced for every plain dense launch (MVD_DENSE_BM / MVD_DENSE_BN,
    read once per process, hence the subprocess): the dense-linear cases above -- ragged M / N / K tails, residual, forced split-K,
    the identity check -- must hold on the new tile shapes too (GEGLU launches keep the 256-row form)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-s", "-k",
                        "test_linear_dense and not four_wave"], cwd=root,
                       env=dict(os.environ, MVD_DENSE_BM="128", MVD_DENSE_BN=str(bn), MVD_PLAN_DEBUG="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]
    assert f"bm=128 bn={bn}" in r.stderr, "the forced 128-row tiles were not used"  # [plan] lines of igemm_go
    worst = max(float(l.split("relL2=")[1].split()[0]) for l in r.stdout.splitlines() if "[parity] dense linear" in l and "relL2=" in l)
    print(f"[parity] dense linear through gemm_dma_kernel<128,{bn}>: worst relL2={worst:.2e}")


def test_linear_geglu(eng):
    M, K, N = 200, 64, 512
    a, w, b = rnd(M, K), rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
    x, gate = F.linear(a, w, b).chunk(2, -1)
    close(eng.op_linear(a, w, b, geglu=True), x * F.gelu(gate), "linear+GEGLU")


@pytest.mark.parametrize("cfg", [
    dict(B=2, Cin=64, H=16, W=16, Cout=96, k=3, stride=1, up=0, res=False, sk=0),
    dict(B=2, Cin=64, H=16, W=16, Cout=96, k=3, stride=2, up=0, res=False, sk=0),
    dict(B=1, Cin=128, H=8, W=8, Cout=64, k=3, stride=1, up=1, res=False, sk=0),
    dict(B=2, Cin=192, H=32, W=32, Cout=64, k=3, stride=1, up=0, res=True, sk=0),
    dict(B=2, Cin=64, H=16, W=16, Cout=128, k=1, stride=1, up=0, res=True, sk=0),
    dict(B=3, Cin=8, H=32, W=32, Cout=64, k=3, stride=1, up=0, res=False, sk=0),
    dict(B=2, Cin=4, H=32, W=32, Cout=16, k=3, stride=1, up=0, res=False, sk=0),
    dict(B=1, Cin=320, H=32, W=32, Cout=4, k=3, stride=1, up=0, res=False, sk=0),
    dict(B=2, Cin=256, H=4, W=4, Cout=256, k=3, stride=1, up=0, res=True, sk=3),
    dict(B=1, Cin=64, H=7, W=5, Cout=40, k=3, stride=2, up=0, res=False, sk=0),
    # LDS-halo 3x3 kernel: 16x16 blocks (32x32 and 16x16 images), 8x8 images (4 per tile), both column widths,
    # partial last tile, split over channel chunks
    dict(B=3, Cin=128, H=32, W=32, Cout=320, k=3, stride=1, up=0, res=True, sk=0),
    dict(B=5, Cin=192, H=16, W=16, Cout=128, k=3, stride=1, up=0, res=False, sk=0),
    dict(B=7, Cin=128, H=8, W=8, Cout=192, k=3, stride=1, up=0, res=True, sk=0),
    dict(B=2, Cin=256, H=16, W=16, Cout=320, k=3, stride=1, up=0, res=True, sk=2),
    dict(B=6, Cin=320, H=8, W=8, Cout=64, k=3, stride=1, up=0, res=False, sk=5),
    # LDS-DMA implicit GEMM with tap gather (fp16 source, M >= 512): stride 2, fused nearest upsample, 4x4 images
    # with split-K, ragged image, 1x1 into a wide layer
    dict(B=4, Cin=64, H=32, W=32, Cout=96, k=3, stride=2, up=0, res=False, sk=0),
    dict(B=2, Cin=128, H=16, W=16, Cout=320, k=3, stride=1, up=1, res=True, sk=0),
    dict(B=32, Cin=256, H=4, W=4, Cout=192, k=3, stride=1, up=0, res=True, sk=0),
    dict(B=20, Cin=64, H=7, W=5, Cout=40, k=3, stride=1, up=0, res=False, sk=4),
    dict(B=3, Cin=192, H=16, W=16, Cout=644, k=1, stride=1, up=0, res=True, sk=0),
    # parity-folded upsample conv (B*H*W >= 2048): even / odd edges, ragged width, residual, split-K
    dict(B=2, Cin=64, H=32, W=32, Cout=96, k=3, stride=1, up=1, res=True, sk=0),
    dict(B=12, Cin=128, H=16, W=12, Cout=72, k=3, stride=1, up=1, res=False, sk=2),
])
def test_conv2d(eng, cfg):
    x = rnd(cfg["B"], cfg["Cin"], cfg["H"], cfg["W"])
    w = rnd(cfg["Cout"], cfg["Cin"], cfg["k"], cfg["k"], seed=3, scale=(cfg["Cin"] * cfg["k"] ** 2) ** -0.5)
    b = rnd(cfg["Cout"], seed=4)
    xin = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if cfg["up"] else x
    want = F.conv2d(xin, w, b, stride=cfg["stride"], padding=cfg["k"] // 2)
    r = rnd(*want.shape, seed=5) if cfg["res"] else None
    if r is not None:
        want = want + r
    got = eng.op_conv(x, w, b, stride=cfg["stride"], upsample=cfg["up"], resid=r, force_splitk=cfg["sk"])
    close(got, want, f"conv2d {cfg}")


@pytest.mark.parametrize("Cin", [32, 64])  # 32: fp32-source gather kernel, 64: fp16-source LDS-DMA kernel
@pytest.mark.parametrize("stride,transposed,res", [(1, False, False), (2, False, False), (1, True, True), (1, True, False)])
def test_conv3d(eng, stride, transposed, res, Cin):
    B, D, H, W, Cout = 2, 6, 8, 8, 32
    x = rnd(B, Cin, D, H, W)
    b = rnd(Cout, seed=4)
    if transposed:
        w = rnd(Cin, Cout, 3, 3, 3, seed=3, scale=(Cin * 27 / 8) ** -0.5)
        want = F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1)
    else:
        w = rnd(Cout, Cin, 3, 3, 3, seed=3, scale=(Cin * 27) ** -0.5)
        want = F.conv3d(x, w, b, stride=stride, padding=1)
    r = rnd(*want.shape, seed=5) if res else None
    if r is not None:
        want = want + r
    close(eng.op_conv3d(x, w, b, stride=stride, transposed=transposed, resid=r), want,
          f"conv3d s={stride} T={transposed} res={res}")


def test_transposed_and_upsample_convs_through_the_parity_walk():
    """Round 6: one workgroup walks the parity classes of its tile (k_gemm.hip PWALK; the planner takes it when the tile grid fills the
    chip by itself: the frustum network's level-0 ConvTranspose3d).  Forced for every parity-batched launch (MVD_PAR_WALK_MIN=1,
    read once per process, hence the subprocess): the ConvTranspose3d cases (8 classes of 1 ... 8 taps, with and without residual)
    and the parity-folded upsample convolutions (4 classes of 4 taps) of this file."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-s", "-k",
                        "(test_conv3d or test_conv2d or test_upconv) and not parity_walk"], cwd=root,
                       env=dict(os.environ, MVD_PAR_WALK_MIN="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]
    worst = max(float(l.split("relL2=")[1].split()[0]) for l in r.stdout.splitlines() if "[parity] conv" in l and "relL2=" in l)
    print(f"[parity] parity-batched convolutions through the parity walk: worst relL2={worst:.2e}")


@pytest.mark.parametrize("B,C,HW,G,eps,act", [(2, 64, 1024, 32, 1e-5, 1), (2, 320, 256, 32, 1e-6, 0), (3, 16, 1024, 8, 1e-5, 1),
                                              (1, 512, 96, 8, 1e-5, 2), (2, 1920, 64, 32, 1e-5, 1), (1, 64, 49152, 8, 1e-5, 1),
                                              (2, 640, 1024, 32, 1e-5, 1), (3, 960, 1024, 32, 1e-5, 1), (2, 1280, 16, 32, 1e-5, 1),
                                              (2, 2560, 64, 32, 1e-5, 1), (2, 1280, 1024, 32, 1e-5, 0), (3, 320, 1024, 32, 1e-5, 1),
                                              (2, 640, 64, 32, 1e-5, 1), (2, 320, 100, 32, 1e-5, 1)])
def test_group_norm(eng, B, C, HW, G, eps, act):
    x = rnd(B, C, HW) * 2.0 + 0.7
    g, b = 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    want = F.group_norm(x, G, g, b, eps)
    want = F.silu(want) if act == 1 else (F.relu(want) if act == 2 else want)
    close(eng.op_group_norm(x, G, g, b, eps, act), want, f"group_norm C={C} HW={HW} G={G} act={act}")


@pytest.mark.parametrize("rows,C", [(257, 320), (64, 1280), (1000, 64), (1030, 640), (33, 1280), (4096, 320)])
def test_layer_norm(eng, rows, C):
    x = rnd(rows, C) * 3.0 - 0.5
    g, b = 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    close(eng.op_layer_norm(x, g, b), F.layer_norm(x, (C,), g, b), f"layer_norm {rows}x{C}")


@pytest.mark.parametrize("B,T,heads,d", [(2, 256, 8, 40), (1, 1024, 8, 40), (2, 64, 8, 80), (3, 16, 8, 160), (1, 1024, 8, 8),
                                         (2, 256, 8, 16), (1, 64, 8, 32), (1, 1024, 2, 160),
                                         # workgroup counts that are not multiples of the 8 XCDs (30 and 6: the bijective
                                         # XCD remap's remainder branch) with a ragged last key / query tile
                                         (3, 200, 5, 40), (1, 130, 3, 80)])
def test_attention(eng, B, T, heads, d):
    C = heads * d
    q, k, v = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2), rnd(B, T, C, seed=3)
    qh, kh, vh = [t.reshape(B, T, heads, d).permute(0, 2, 1, 3) for t in (q, k, v)]
    want = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh
    want = want.permute(0, 2, 1, 3).reshape(B, T, C)
    close(eng.op_attention(q, k, v, heads), want, f"attention B={B} T={T} d={d}", rel=2e-3, mx=6e-3)


def test_attention_peaked_rows(eng):
    """Forces the online-softmax rescale branch: one key dominates late in the sequence."""
    B, T, heads, d = 1, 256, 8, 40
    C = heads * d
    q, k, v = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2), rnd(B, T, C, seed=3)
    k[:, 200] = q[:, 7] * 4.0
    qh, kh, vh = [t.reshape(B, T, heads, d).permute(0, 2, 1, 3) for t in (q, k, v)]
    want = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, T, C)
    close(eng.op_attention(q, k, v, heads), want, "attention peaked", rel=2e-3, mx=6e-3)


# ---- row-chain kernel (k_rowchain.hip): [to_out + t0] -> LayerNorm3 -> FF1 -> GEGLU -> FF2 -> + t2 [-> proj_out + x_in] in one launch
def _st_tail_case(C, rows, T, ao, po, seed=0):
    g = torch.Generator().manual_seed(1000 * C + rows + 7 * seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    d = dict(xin=r(rows, C), ln_g=1.0 + 0.2 * r(C), ln_b=0.2 * r(C), w1=r(8 * C, C, sc=C ** -0.5), b1=0.3 * r(8 * C),
             w2=r(C, 4 * C, sc=(4 * C) ** -0.5), b2=0.3 * r(C))
    if ao:
        d.update(ao=r(rows, C), w_ao=r(C, C, sc=C ** -0.5), b_ao=0.3 * r(C), rowbias=0.5 * r(rows // T, C), T=T)
    if po:
        d.update(w_po=r(C, C, sc=C ** -0.5), b_po=0.3 * r(C), resid=r(rows, C))
    return d


def _st_tail_ref(d):
    t2 = d["xin"].double()
    if "ao" in d:
        T = d["T"]
        a = d["ao"].half().double()  # the attention output is an fp16 tensor in the engine
        t2 = t2 + a @ d["w_ao"].double().t() + d["b_ao"].double() + d["rowbias"].double().repeat_interleave(T, 0)
    x = F.layer_norm(t2, (t2.shape[1],), d["ln_g"].double(), d["ln_b"].double(), 1e-5)
    hmid = x @ d["w1"].double().t() + d["b1"].double()
    v, gate = hmid.chunk(2, -1)
    t3 = t2 + (v * F.gelu(gate)) @ d["w2"].double().t() + d["b2"].double()
    if "w_po" in d:
        return (t3 @ d["w_po"].double().t() + d["b_po"].double() + d["resid"].double()).float()
    return t3.float()


@pytest.mark.parametrize("C,rows,T,ao,po", [
    (64, 256, 64, False, False), (64, 1024, 1024, True, True), (128, 512, 256, True, False), (128, 256, 32, True, True),
    (256, 384, 64, True, True), (320, 1024, 1024, False, False), (320, 2048, 1024, True, True), (320, 4096, 1024, True, False),
    (64, 128, 128, True, True),
])
def test_st_tail_rowchain(eng, C, rows, T, ao, po):
    d = _st_tail_case(C, rows, T, ao, po)
    got = eng.op_st_tail(**d)
    close(got, _st_tail_ref(d), f"st_tail C={C} rows={rows} ao={ao} po={po}")


def test_st_tail_rowchain_split_output(eng):
    """fp16 result written as [hi | lo | hi] rows (the operand of an extended-precision proj_out): hi + lo carries ~22 bits."""
    d = _st_tail_case(320, 1024, 1024, True, False, seed=3)
    got = eng.op_st_tail(split=True, **d)
    close(got, _st_tail_ref(d), "st_tail split", rel=6e-4, mx=3e-3)


def test_st_tail_rowchain_identity_weights(eng):
    """Transposition / permutation check: identity-like projections with asymmetric FF weights must reproduce the reference
    (a swapped k permutation or row block shows as O(1) error, not as rounding)."""
    C, rows = 128, 256
    d = _st_tail_case(C, rows, 256, True, True, seed=5)
    d["w_ao"] = torch.eye(C) + 0.01 * torch.arange(C * C, dtype=torch.float32).reshape(C, C).remainder(7.0) / 7.0
    d["w_po"] = torch.eye(C).roll(3, 0)
    close(eng.op_st_tail(**d), _st_tail_ref(d), "st_tail identity")


# ---- conv3x (k_conv3x.hip): 3x3 stride-1 convolutions at 16-divisible resolutions, weights as a pre-packed fragment stream
@pytest.mark.parametrize("B,Cin,H,W,Cout,res,sk", [
    (1, 64, 16, 16, 160, False, 0), (2, 128, 32, 32, 320, True, 0), (3, 192, 16, 32, 128, True, 0), (2, 320, 16, 16, 640, True, 2),
    (1, 960, 32, 32, 320, False, 0), (2, 64, 48, 16, 256, True, 0), (1, 256, 16, 16, 160, False, 4),
    # 8 x 8 images: four per workgroup tile (a partial last tile at B = 6), split over the channel chunks
    (8, 128, 8, 8, 160, True, 0), (6, 192, 8, 8, 320, True, 0), (4, 1280, 8, 8, 1280, True, 4), (1, 64, 8, 8, 128, False, 0),
    # 8 pixel tiles x 8 column tiles of 3.7 MB of weights each: walked column-tile-major (xcd_prefers_cols)
    (32, 1280, 8, 8, 1280, True, 0), (30, 640, 8, 8, 1280, False, 2),
])
def test_conv3x(eng, B, Cin, H, W, Cout, res, sk):
    x = rnd(B, Cin, H, W)
    w = rnd(Cout, Cin, 3, 3, seed=3, scale=(Cin * 9) ** -0.5)
    b = rnd(Cout, seed=4)
    want = F.conv2d(x, w, b, padding=1)
    r = rnd(*want.shape, seed=5) if res else None
    if r is not None:
        want = want + r
    close(eng.op_conv(x, w, b, resid=r, force_splitk=sk), want, f"conv3x B={B} {Cin}->{Cout} {H}x{W} res={res} sk={sk}")


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(4, 8, 32, 32, 320), (3, 8, 16, 16, 64), (2, 4, 12, 20, 96), (1, 8, 6, 6, 40), (2, 4, 5, 10, 128)])
def test_first_layer_conv_exact_fp32(eng, B, Cin, H, W, Cout):
    """The UNet's first 3x3 convolution at inference (k_misc.hip: conv_in_f32_kernel): fp32 multiply-adds, so it matches torch's
    fp32 convolution to summation order (reference: ldm/modules/diffusionmodules/openaimodel.py:606-610, conv_nd(dims, in_channels,
    model_channels, 3, padding=1))."""
    torch.manual_seed(5)
    x, w, b = torch.randn(B, Cin, H, W), torch.randn(Cout, Cin, 3, 3) * 0.2, torch.randn(Cout)
    got = eng.op_conv(x, w, b, force_splitk=-2)
    close(got, F.conv2d(x, w, b, padding=1), f"first-layer conv {Cin}->{Cout} {H}x{W}", rel=2e-6, mx=1e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(4, 320, 32, 32, 4), (2, 64, 16, 16, 4), (1, 96, 8, 48, 3), (3, 32, 5, 16, 1)])
def test_output_head_conv_exact_fp32(eng, B, Cin, H, W, Cout):
    """The UNet's last 3x3 convolution at inference (k_misc.hip: out_conv_f32_kernel; reference:
    ldm/modules/diffusionmodules/openaimodel.py:717-721, zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1))):
    fp32 multiply-adds in a fixed order."""
    torch.manual_seed(6)
    x, w, b = torch.randn(B, Cin, H, W), torch.randn(Cout, Cin, 3, 3) * 0.05, torch.randn(Cout)
    got = eng.op_conv(x, w, b, force_splitk=-3)
    close(got, F.conv2d(x, w, b, padding=1), f"output-head conv {Cin}->{Cout} {H}x{W}", rel=3e-6, mx=1e-5)


def test_conv3x_impulse(eng):
    """A one-hot input pixel / channel reproduces the (flipped) kernel around it: catches a swapped tap, a transposed fragment or a
    mis-placed halo row exactly (products are exact in fp16 for these weights)."""
    Cin, Cout, H = 64, 160, 16
    w = (torch.arange(Cout * Cin * 9, dtype=torch.float32).reshape(Cout, Cin, 3, 3) % 127 - 63) / 64.0
    x = torch.zeros(1, Cin, H, H)
    x[0, 5, 7, 9] = 1.0
    x[0, 63, 0, 15] = 2.0
    close(eng.op_conv(x, w), F.conv2d(x, w, padding=1), "conv3x impulse", rel=1e-6, mx=1e-6)


# ---- row-head kernel (k_rowchain.hip): proj_in -> t0, LayerNorm1, q | k | v in one launch
@pytest.mark.parametrize("rows", [128, 1024, 4096])
def test_st_head_rowhead(eng, rows):
    C = 320
    g = torch.Generator().manual_seed(77 + rows)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    n0, w_pi, b_pi = r(rows, C), r(C, C, sc=C ** -0.5), 0.3 * r(C)
    ln_g, ln_b = 1.0 + 0.2 * r(C), 0.2 * r(C)
    wq, wk, wv = (r(C, C, sc=C ** -0.5) for _ in range(3))
    t0, qkv = eng.op_st_head(n0, w_pi, b_pi, ln_g, ln_b, wq, wk, wv)
    t0_ref = n0.half().double() @ w_pi.double().t() + b_pi.double()
    l1 = F.layer_norm(t0_ref, (C,), ln_g.double(), ln_b.double(), 1e-5)
    qkv_ref = l1 @ torch.cat([wq, wk, wv]).double().t()
    close(t0, t0_ref.float(), f"st_head t0 rows={rows}")
    close(qkv, qkv_ref.float(), f"st_head qkv rows={rows}")


def test_st_head_rowhead_asymmetric(eng):
    """A permutation as proj_in and distinct q / k / v row patterns: a swapped row block or k permutation shows as O(1) error."""
    C, rows = 320, 256
    n0 = (torch.arange(rows * C, dtype=torch.float32).reshape(rows, C) % 97 - 48) / 32.0
    w_pi = torch.eye(C).roll(5, 0)
    wq = torch.eye(C).roll(1, 1) * 0.5
    wk = torch.diag(torch.linspace(0.5, 1.5, C))
    wv = torch.eye(C).flip(0)
    ln_g, ln_b = torch.linspace(0.8, 1.2, C), torch.linspace(-0.1, 0.1, C)
    t0, qkv = eng.op_st_head(n0, w_pi, torch.zeros(C), ln_g, ln_b, wq, wk, wv)
    t0_ref = n0.double() @ w_pi.double().t()
    l1 = F.layer_norm(t0_ref, (C,), ln_g.double(), ln_b.double(), 1e-5)
    close(t0, t0_ref.float(), "st_head asym t0", rel=1e-6, mx=1e-6)
    close(qkv, (l1 @ torch.cat([wq, wk, wv]).double().t()).float(), "st_head asym qkv")


def test_rowchain_and_rowhead_extended_precision_forms(eng):
    """The forms the last output block takes at the default precision level: proj_out / proj_in with both operands split into
    fp16 hi + lo parts inside the kernels (three products).  Against fp64 the extended-precision projection must beat the fp16
    floor of the plain form by a wide margin on t0 (one GEMM deep), and the whole tail must stay within the bound."""
    C, rows, T = 320, 2048, 1024
    d = _st_tail_case(C, rows, T, True, True, seed=9)
    close(eng.op_st_tail(xp_out=True, **d), _st_tail_ref(d), "st_tail xp proj_out")
    g = torch.Generator().manual_seed(91)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    n0, w_pi, b_pi = r(rows, C), r(C, C, sc=C ** -0.5), 0.3 * r(C)
    ln_g, ln_b = 1.0 + 0.2 * r(C), 0.2 * r(C)
    wq, wk, wv = (r(C, C, sc=C ** -0.5) for _ in range(3))
    t0, qkv = eng.op_st_head(n0, w_pi, b_pi, ln_g, ln_b, wq, wk, wv, xp=True)
    t0_ref = n0.double() @ w_pi.double().t() + b_pi.double()   # n0 is NOT rounded to fp16 in this form
    l1 = F.layer_norm(t0_ref, (C,), ln_g.double(), ln_b.double(), 1e-5)
    close(t0, t0_ref.float(), "st_head xp t0", rel=5e-6, mx=5e-5)
    close(qkv, (l1 @ torch.cat([wq, wk, wv]).double().t()).float(), "st_head xp qkv")
