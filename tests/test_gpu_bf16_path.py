"""The bf16 data path (BASELINE configs[2-4], ops.set_conv_precision("bf16")): activations are bf16 tensors in HBM between
our own layers, every kernel computes in fp32 registers.  Each op is held to its fp32 counterpart on the SAME
(bf16-representable) inputs: the bf16 result must be the fp32 result rounded once -- relative error <= 2^-8 per element
(plus 2^-8 more per additional rounding where a backward consumes a rounded gradient)."""
import pytest
import torch
import torch.nn.functional as F

from c2m_amd import ops
from gpu_util import rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 2.0 ** -8


def bf(t):
    return t.bfloat16().float()


def g(t):
    return t.to(DEV)


def near(got, ref, roundings=1, what=""):
    """elementwise |got - ref| <= roundings * 2^-8 * |ref| + tiny absolute floor tied to the tensor scale"""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    tol = roundings * EPS * ref.abs() + 2 * EPS * EPS * float(ref.abs().max()) + 1e-30
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements off; worst {float(((got - ref).abs() / (ref.abs() + 1e-12))[bad].max()):.3e}"


@pytest.mark.parametrize("case", [((2, 48, 16, 32), 64, 3, 1, 1, "reflect"), ((1, 130, 24, 64), 136, 3, 1, 1, "zeros"),
                                  ((2, 21, 16, 32), 64, 4, 2, 1, "reflect"), ((2, 32, 5, 16, 32), 40, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),
                                  ((8, 512, 2, 4), 512, 3, 1, 1, "reflect"), ((1, 32, 128, 160), 2, 3, 1, 1, "reflect")])
def test_conv_bf16_tensors_in_and_out(case):
    """bf16 X in, bf16 Y out (fp32 for the <= 4-channel head), bf16 dY in, bf16 dX out, fp32 dW: against the fp32 convolution
    of the same values.  Covers the LDS-patch kernel, the gather kernel (strided, 3-D), split-K and the thin head."""
    xs, cout, k, stride, pad, mode = case
    nd = len(xs) - 2
    kk = (k,) * nd if isinstance(k, int) else k
    x = bf(rnd(1, *xs))
    w = bf(rnd(2, cout, xs[1], *kk, scale=(1.0 / (xs[1] * int(torch.tensor(kk).prod()))) ** 0.5))
    b = rnd(3, cout, scale=0.1)
    with ops.conv_precision("fp32"):
        xr, wr, br = (g(t).requires_grad_(True) for t in (x, w, b))
        yr = ops.conv(xr, wr, br, stride=stride, padding=pad, padding_mode=mode, act="lrelu")
        go = bf(rnd(4, *yr.shape))
        (yr * g(go)).sum().backward()
    with ops.conv_precision("bf16"):
        xb = g(x).bfloat16().requires_grad_(True)
        wb, bb = g(w).requires_grad_(True), g(b).requires_grad_(True)
        y = ops.conv(xb, wb, bb, stride=stride, padding=pad, padding_mode=mode, act="lrelu")
        assert y.dtype == (torch.bfloat16 if cout > 4 else torch.float32)
        y.backward(g(go).to(y.dtype))
    near(y, yr, 1, "conv fwd")
    assert xb.grad.dtype == torch.bfloat16 and wb.grad.dtype == torch.float32
    # the backward starts from the bf16-rounded y (activation mask) and the same dY: dX is one more rounding away
    scale = float(xr.grad.abs().max())
    assert float((xb.grad.float() - xr.grad).abs().max()) <= 3 * EPS * scale, "conv dgrad"
    wscale = float(wr.grad.abs().max())
    assert float((wb.grad - wr.grad).abs().max()) <= 2e-3 * wscale, "conv wgrad (fp32 accumulation of exact products)"
    assert float((bb.grad - br.grad).abs().max()) <= 2e-3 * float(br.grad.abs().max()) + 1e-4


@pytest.mark.parametrize("shape,mode", [((4, 16, 32, 64), "bn"), ((3, 8, 5, 16, 32), "bn"), ((2, 12, 24, 40), "in"), ((5, 6, 3, 7), "in")])
def test_norm_act_bf16_io(shape, mode):
    x = bf(rnd(11, *shape) * 1.5 + 0.3)
    C = shape[1]
    gam, bet = rnd(12, C) * 0.2 + 1.0, rnd(13, C) * 0.1
    go = bf(rnd(14, *shape))

    def run(dtype):
        xg = g(x).to(dtype).requires_grad_(True)
        gg, bg = g(gam).requires_grad_(True), g(bet).requires_grad_(True)
        if mode == "bn":
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            y = ops.batch_norm_act(xg, gg, bg, rm, rv, act="lrelu")
        else:
            rm = None
            y = ops.instance_norm_act(xg, gg, bg, act="lrelu")
        y.backward(g(go).to(dtype))
        return y, xg.grad, gg.grad, bg.grad, rm

    yr, dxr, dgr, dbr, rmr = run(torch.float32)
    y, dx, dg, db, rm = run(torch.bfloat16)
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dg.dtype == torch.float32
    near(y, yr, 1, "norm fwd")
    assert float((dx.float() - dxr).abs().max()) <= 3 * EPS * float(dxr.abs().max()), "norm dx"
    assert torch.allclose(dg, dgr, rtol=1e-4, atol=1e-4) and torch.allclose(db, dbr, rtol=1e-4, atol=1e-4)
    if rm is not None:
        assert torch.equal(rm, rmr), "statistics are computed in fp32 from the same values"


def test_spade_norm_bf16_io():
    x, gb_ = bf(rnd(21, 2, 6, 16, 72) + 0.5), bf(0.3 * rnd(22, 2, 12, 16, 72))
    go = bf(rnd(23, 2, 6, 16, 72))

    def run(dtype):
        xg, gg = g(x).to(dtype).requires_grad_(True), g(gb_).to(dtype).requires_grad_(True)
        y = ops.spade_norm_act(xg, gg, "lrelu")
        y.backward(g(go).to(dtype))
        return y, xg.grad, gg.grad

    yr, dxr, dgr = run(torch.float32)
    y, dx, dgb = run(torch.bfloat16)
    assert y.dtype == dx.dtype == dgb.dtype == torch.bfloat16
    near(y, yr, 1, "spade fwd")
    near(dgb, dgr, 1, "spade d(gamma, beta) maps")
    assert float((dx.float() - dxr).abs().max()) <= 3 * EPS * float(dxr.abs().max())


def test_elementwise_ops_bf16_io():
    """upsample x2, max-pool, resize, flow warp (+ occlusion), L1: bf16 tensors in and out == the fp32 op rounded once."""
    x = bf(rnd(31, 3, 5, 16, 32))
    go2 = bf(rnd(32, 3, 5, 32, 64))
    for name, fn, go in (("upsample2x", ops.upsample2x, go2), ("maxpool", ops.maxpool2x2, bf(rnd(33, 3, 5, 8, 16)))):
        xr = g(x).requires_grad_(True)
        yr = fn(xr)
        yr.backward(g(go))
        xb = g(x).bfloat16().requires_grad_(True)
        y = fn(xb)
        y.backward(g(go).bfloat16())
        assert y.dtype == torch.bfloat16 and xb.grad.dtype == torch.bfloat16
        near(y, yr, 1, name)
        near(xb.grad, xr.grad, 1, name + " backward")
    near(ops.resize_bilinear(g(x).bfloat16(), (8, 20), align_corners=True), ops.resize_bilinear(g(x), (8, 20), align_corners=True), 1, "resize")
    flow, occ = rnd(34, 3, 2, 16, 32, scale=2.5), torch.rand(3, 1, 16, 32, generator=torch.Generator().manual_seed(35))
    go = bf(rnd(36, 3, 5, 16, 32))
    ir, fr = g(x).requires_grad_(True), g(flow).requires_grad_(True)
    yr = ops.flow_warp(ir, fr, g(occ))
    yr.backward(g(go))
    ib, fb = g(x).bfloat16().requires_grad_(True), g(flow).requires_grad_(True)
    y = ops.flow_warp(ib, fb, g(occ))
    y.backward(g(go).bfloat16())
    assert y.dtype == torch.bfloat16 and ib.grad.dtype == torch.bfloat16 and fb.grad.dtype == torch.float32
    near(y, yr, 1, "flow_warp")
    assert float((ib.grad.float() - ir.grad).abs().max()) <= 2 * EPS * float(ir.grad.abs().max())
    assert torch.equal(fb.grad, fr.grad), "d(flow) is accumulated in fp32 from identical values"
    a, b = bf(rnd(37, 2, 7, 9, 11)), bf(rnd(38, 2, 7, 9, 11))
    ar = g(a).requires_grad_(True)
    lr = ops.l1_mean(ar, g(b))
    lr.backward()
    ab = g(a).bfloat16().requires_grad_(True)
    lb = ops.l1_mean(ab, g(b).bfloat16())
    lb.backward()
    assert lb.dtype == torch.float32 and torch.equal(lb, lr) and ab.grad.dtype == torch.bfloat16
    near(ab.grad, ar.grad, 1, "l1 backward")


def test_model_activations_are_bf16_in_bf16_mode():
    """The point of the path: between our own layers the tensors really are bf16 (not fp32 with rounded operands)."""
    from c2m_amd.modules.layers.same_block import SameBlock2d
    from c2m_amd.modules.layers.down_block import DownBlock2d
    from c2m_amd.modules.layers.up_block import UpBlock2d
    torch.manual_seed(0)
    a, d, u = SameBlock2d(16, 32).to(DEV), DownBlock2d(32, 64, kernel_size=4, stride=2, padding=1, padding_mode="reflect").to(DEV), UpBlock2d(64, 32, reshape_3d=False, input_2d=True).to(DEV)
    x = torch.randn(5, 16, 32, 64, device=DEV)
    with ops.conv_precision("bf16"):
        h1 = a(x)
        h2 = d(h1)
        h3 = u(h2)
    assert h1.dtype == h2.dtype == h3.dtype == torch.bfloat16
    (h3.float().square().mean()).backward()
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all())
               for m in (a, d, u) for p in m.parameters())
