"""GPU: the backward kernels of the training step (k_bwd.hip) one by one, through the C ABI, against torch.autograd of the
same op in fp32 on the CPU.  Bounds: fp32 kernels 1e-5; the attention backward computes on fp16 operands (its inputs are
rounded to fp16 on both sides of the comparison, what remains is the fp16 rounding of P / dS inside the kernel): 3e-3."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=2.0)
    yield e
    e.close()


def rel(a, b):
    return ((a.cpu() - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("B,T,heads,d", [(2, 1024, 8, 40), (2, 256, 8, 80), (3, 64, 8, 160), (2, 1024, 8, 8), (1, 256, 8, 16),
                                         (2, 64, 8, 32), (1, 4096, 2, 64)])
def test_attention_backward(eng, B, T, heads, d):
    g = torch.Generator().manual_seed(T + d)
    C = heads * d
    q, k, v, do = [torch.randn(B, T, C, generator=g).half().float() for _ in range(4)]
    q *= 1.5  # sharper softmax than unit-variance scores
    qq, kk, vv = [t.clone().requires_grad_(True) for t in (q, k, v)]

    def split(t):
        return t.view(B, T, heads, d).transpose(1, 2)

    att = torch.softmax(split(qq) @ split(kk).transpose(-1, -2) / d ** 0.5, -1)
    out = (att @ split(vv)).transpose(1, 2).reshape(B, T, C)
    out.backward(do)
    dq, dk, dv = eng.op_attention_bwd(q, k, v, do, heads)
    errs = [rel(dq, qq.grad), rel(dk, kk.grad), rel(dv, vv.grad)]
    print(f"[parity] attention backward B={B} T={T} d={d}: dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}")
    assert max(errs) <= 3e-3, errs


@pytest.mark.parametrize("B,rows,C,G,act", [(2, 1024, 320, 32, 1), (3, 256, 960, 32, 1), (2, 64, 2560, 32, 1), (2, 16, 1280, 32, 0),
                                            (2, 1024, 128, 8, 2), (1, 49152, 64, 8, 2), (2, 64, 1024, 8, 1)])
def test_group_norm_backward(eng, B, rows, C, G, act):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(B, rows, C, generator=g) * 1.3 + 0.2
    dy = torch.randn(B, rows, C, generator=g)
    gamma, beta = torch.randn(C, generator=g) * 0.3 + 1.0, torch.randn(C, generator=g) * 0.2
    xx, gg, bb = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.nn.functional.group_norm(xx.transpose(1, 2), G, gg, bb, 1e-5).transpose(1, 2)
    y = torch.nn.functional.silu(y) if act == 1 else (torch.relu(y) if act == 2 else y)
    y.backward(dy)
    dx, dg, db = eng.op_group_norm_bwd(x, dy, G, gamma, beta, 1e-5, act)
    errs = [rel(dx, xx.grad), rel(dg, gg.grad), rel(db, bb.grad)]
    print(f"[parity] GroupNorm backward rows={rows} C={C} G={G} act={act}: dx {errs[0]:.2e} dgamma {errs[1]:.2e} dbeta {errs[2]:.2e}")
    assert max(errs) <= 2e-5, errs


@pytest.mark.parametrize("rows,C", [(2048, 320), (512, 640), (130, 1280), (7, 64)])
def test_layer_norm_backward(eng, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2.0 - 0.5
    dy = torch.randn(rows, C, generator=g)
    gamma, beta = torch.randn(C, generator=g) * 0.3 + 1.0, torch.randn(C, generator=g) * 0.2
    xx, gg, bb = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xx, (C,), gg, bb, 1e-5).backward(dy)
    dx, dg, db = eng.op_layer_norm_bwd(x, dy, gamma)
    errs = [rel(dx, xx.grad), rel(dg, gg.grad), rel(db, bb.grad)]
    print(f"[parity] LayerNorm backward rows={rows} C={C}: dx {errs[0]:.2e} dgamma {errs[1]:.2e} dbeta {errs[2]:.2e}")
    assert max(errs) <= 2e-5, errs
