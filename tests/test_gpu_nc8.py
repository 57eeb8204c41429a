"""Channel-blocked (NC8) bf16 convolution kernels of round 4 (c2m_amd/csrc/conv_nc8.hip) against an fp32 convolution of the same
bf16-representable operands (every product exact in fp32: only the summation order differs) -- the layout pass, every tile /
buffering variant of the 3x3 patch kernel, the stride-2 parity form, the transposed-read weight gradient; ragged tiles, channel
counts off the 8 / 16 / 64 grids, split-K, zeros and reflect padding, NaNs planted behind the tensors.  Every case asserts that the
plan really routes to the NC8 kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from c2m_amd import ops
from gpu_util import rel_close, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _three_launch_norms():
    """The tests of this file compare hand-over MODES (NCHW tensors against NC8-only tensors) bit for bit.  The NCHW hand-over is
    eligible for the one-launch norm kernels of round 5 (another summation order than the three-launch path, which is the only one
    that can write an NC8-only result): bit-identity is a statement about the LAYOUT, so every mode runs the same norm here."""
    from c2m_amd import _lib
    old = _lib.lib().c2m_norm_set_fused(0)
    yield
    _lib.lib().c2m_norm_set_fused(old)
DEV = "cuda:0"


def _bf(t):
    return t.bfloat16().float()


@pytest.fixture(autouse=True)
def _bf16_mode():
    prev = ops.set_conv_precision("bf16")
    v0, p0, f0, d0 = ops._NC8_VARIANT, ops._NC8_S2_WGRAD_MIN_PIX, ops._NC8_S2_FILL_G8, ops._NC8_DGRAD_FILL
    ops._NC8_S2_WGRAD_MIN_PIX = 0          # (the small test maps would otherwise stay on the NCHW weight-gradient kernel)
    # (half-filled stride-2 tiles and padded data-gradient domains stay on the patch forms here -- their two-target / ragged-row
    # epilogues are what these cases are for; the gather form has tests/test_gpu_g8.py)
    ops._NC8_S2_FILL_G8 = ops._NC8_DGRAD_FILL = ops._NC8_FILL
    ops._geom_cache.clear()
    yield
    ops._NC8_VARIANT, ops._NC8_S2_WGRAD_MIN_PIX, ops._NC8_S2_FILL_G8, ops._NC8_DGRAD_FILL = v0, p0, f0, d0
    ops._geom_cache.clear()
    ops.set_conv_precision(prev)


@pytest.mark.parametrize("N,C,H,W", [(2, 8, 8, 16), (3, 13, 6, 20), (1, 64, 16, 32), (2, 3, 4, 8)])
def test_nchw_to_nc8_layout(N, C, H, W):
    x = rnd(1, N, C, H, W).to(DEV).bfloat16()
    got = ops._to_nc8(x)
    CB = (C + 7) // 8
    ref = torch.zeros(N, CB * 8, H, W, device=DEV, dtype=torch.bfloat16)
    ref[:, :C] = x
    ref = ref.view(N, CB, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    assert got.shape == (N, CB, H, W, 8) and torch.equal(got, ref)


def _ref(x, w, b, stride, mode):
    pad = 1
    xp = F.pad(x, (pad,) * 4, mode="reflect") if mode == "reflect" else F.pad(x, (pad,) * 4)
    return F.conv2d(xp, w, b, stride=stride)


PATCH_CASES = [   # N, Cin, H, W, Cout, mode
    (2, 32, 16, 32, 64, "zeros"), (1, 40, 24, 64, 200, "reflect"), (2, 16, 20, 40, 32, "reflect"), (1, 72, 36, 96, 24, "zeros"),
    (3, 28, 12, 40, 136, "reflect"), (1, 128, 32, 64, 128, "zeros"),
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_patch_nc8_forward_backward(case, variant):
    N, Cin, H, W, Cout, mode = case
    if (variant in (4, 5)) != (Cout <= 32) and variant != 0:
        pytest.skip("32-row variants serve <= 32 output channels, the others more")
    ops._NC8_VARIANT = variant
    ops._geom_cache.clear()
    x, w = _bf(rnd(11, N, Cin, H, W)), _bf(rnd(12, Cout, Cin, 3, 3, scale=(1.0 / (Cin * 9)) ** 0.5))
    b = rnd(13, Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = _ref(xr, wr, br, 1, mode)
    go = _bf(rnd(14, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    pl = ops._plan(xg.bfloat16(), wg, (1, 1, 1), (0, 1, 1), mode == "reflect")
    assert pl.nc8 and pl.fwd_patch, "the forward must run on the NC8 patch kernel"
    # (the data gradient reduces over Cout: 24 output channels would pad a 16-channel chunk by a third -> gather kernel)
    assert any(c["patch"] for c in pl.classes) == (Cout != 24), "data gradient on the NC8 patch kernel"
    y = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode)
    (y * go.to(DEV)).sum().backward()
    assert y.dtype == torch.bfloat16
    rel_close(y.float(), yr, 4e-3, "NC8 forward (one RNE rounding of the exact sum)")
    rel_close(xg.grad, xr.grad, 5e-5, "NC8 data gradient")
    rel_close(wg.grad, wr.grad, 1e-4, "weight gradient")
    rel_close(bg.grad, br.grad, 1e-4, "bias gradient")
    assert pl.wgrad_nc8 == (Cout >= 64 and Cin >= 16)


def test_patch_nc8_split_k_and_deep_reduction():
    """A 512-deep layer on a 8x32 map: few pixel tiles -> split-K slabs + c2m_splitk_reduce; and the 16-row-tile rule."""
    N, Cin, H, W, Cout = 2, 512, 8, 32, 64
    x, w = _bf(rnd(21, N, Cin, H, W)), _bf(rnd(22, Cout, Cin, 3, 3, scale=(1.0 / (Cin * 9)) ** 0.5))
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    pl = ops._plan(xg.bfloat16(), wg, (1, 1, 1), (0, 1, 1), False)
    assert pl.nc8 and pl.fwd_patch and pl.fwd_splits > 1, "split-K case"
    y = ops.conv(xg, wg, None, stride=1, padding=1)
    rel_close(y.float(), F.conv2d(x, w, None, padding=1), 4e-3, "NC8 forward through split-K slabs")


WGRAD_CASES = [(2, 16, 12, 40, 64, "zeros"), (1, 40, 9, 72, 72, "reflect"), (3, 64, 8, 32, 128, "reflect"), (1, 24, 21, 64, 200, "zeros")]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_wgrad_nc8_ragged_chunks(case):
    """Weight gradient from NC8 operands where the 4 x 32-pixel chunks hang over the map (H % 4 != 0, W % 32 != 0) and the channel
    tiles over Cout / Cin (not multiples of 64 / 32 / 8)."""
    N, Cin, H, W, Cout, mode = case
    x, w = _bf(rnd(31, N, Cin, H, W)), _bf(rnd(32, Cout, Cin, 3, 3, scale=(1.0 / (Cin * 9)) ** 0.5))
    b = rnd(33, Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = _ref(xr, wr, br, 1, mode)
    go = _bf(rnd(34, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    pl = ops._plan(xg.bfloat16(), wg, (1, 1, 1), (0, 1, 1), mode == "reflect")
    assert pl.wgrad_nc8, "the case must run on conv_wgrad_nc8_kernel"
    y = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode)
    (y * go.to(DEV)).sum().backward()
    rel_close(wg.grad, wr.grad, 1e-4, "NC8 weight gradient")
    rel_close(bg.grad, br.grad, 1e-4, "NC8 bias gradient")
    rel_close(xg.grad, xr.grad, 5e-5, "data gradient")
    # bit-repeatable (fixed-order slab reduction, no atomics)
    wg2, xg2 = w.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    y2 = ops.conv(xg2, wg2, bg.detach(), stride=1, padding=1, padding_mode=mode)
    (y2 * go.to(DEV)).sum().backward()
    assert torch.equal(wg2.grad, wg.grad)


S2_CASES = [(2, 16, 32, 128, 64, "reflect"), (1, 40, 24, 96, 100, "zeros"), (3, 64, 24, 128, 32, "reflect"), (1, 32, 36, 72, 136, "zeros")]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_stride2_parity_form_forward_and_data_gradient(case):
    """4x4 stride-2 pad-1: forward on the input-parity form, data gradient as four output-parity classes over one dY patch."""
    N, Cin, H, W, Cout, mode = case
    x, w = _bf(rnd(41, N, Cin, H, W)), _bf(rnd(42, Cout, Cin, 4, 4, scale=(1.0 / (Cin * 16)) ** 0.5))
    b = rnd(43, Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = _ref(xr, wr, br, 2, mode)
    go = _bf(rnd(44, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    pl = ops._plan(xg.bfloat16(), wg, (1, 2, 2), (0, 1, 1), mode == "reflect")
    assert pl.s2_nc8 and pl.s2_dgrad_nc8, "the case must run on the stride-2 forms of the NC8 kernels"
    assert pl.s2_wgrad_nc8 == (Cout >= 64 and Cin >= 16), "weight gradient on the stride-2 transposed-read kernel"
    y = ops.conv(xg, wg, bg, stride=2, padding=1, padding_mode=mode)
    (y.float() * go.to(DEV)).sum().backward()
    rel_close(y.float(), yr, 4e-3, "stride-2 NC8 forward")
    rel_close(xg.grad, xr.grad, 5e-5, "stride-2 NC8 data gradient")
    rel_close(wg.grad, wr.grad, 1e-4, "weight gradient")
    rel_close(bg.grad, br.grad, 1e-4, "bias gradient")
    # bf16 target of the data gradient (the type it has inside the network)
    xb = x.to(DEV).bfloat16().requires_grad_(True)
    yb = ops.conv(xb, wg.detach(), None, stride=2, padding=1, padding_mode=mode)
    yb.backward(go.to(DEV).bfloat16())
    assert xb.grad.dtype == torch.bfloat16
    # (reflect: the padded gradient is rounded to bf16, then the fold adds up to four of those values and rounds again)
    rel_close(xb.grad.float(), xr.grad, 8e-3 if mode == "reflect" else 4e-3, "stride-2 NC8 data gradient, bf16 result")


def test_nc8_kernels_never_read_past_their_inputs():
    """The DMAs carry the channel block in an offset the hardware does not range-check: an odd number of 8-channel blocks (last
    16-channel chunk half empty) must read zeros, not the NaNs planted behind the tensors."""
    N, Cin, Cout, H, W = 1, 40, 72, 16, 64
    pool = torch.full((2 * N * Cin * H * W + 8192,), float("nan"), device=DEV, dtype=torch.bfloat16)
    x = pool[:N * Cin * H * W].view(N, Cin, H, W)
    x.copy_(_bf(rnd(51, N, Cin, H, W)).to(DEV))
    gpool = torch.full((2 * N * Cout * H * W + 8192,), float("nan"), device=DEV, dtype=torch.bfloat16)
    go = gpool[:N * Cout * H * W].view(N, Cout, H, W)
    go.copy_(_bf(rnd(52, N, Cout, H, W)).to(DEV))
    w = _bf(rnd(53, Cout, Cin, 3, 3, scale=(1.0 / (Cin * 9)) ** 0.5)).to(DEV).requires_grad_(True)
    xg = x.requires_grad_(True)
    y = ops.conv(xg, w, None, stride=1, padding=1)
    y.backward(go)
    for t in (y, xg.grad, w.grad):
        assert bool(torch.isfinite(t.float()).all())
    xr, wr = x.detach().float().cpu().requires_grad_(True), w.detach().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=1)
    yr.backward(go.float().cpu())
    rel_close(y.float(), yr, 4e-3, "forward")
    rel_close(xg.grad.float(), xr.grad, 4e-3, "data gradient (bf16 result)")
    rel_close(w.grad, wr.grad, 1e-4, "weight gradient")


K333_CASES = [(1, 32, 3, 16, 64, 64, "reflect"), (2, 40, 5, 24, 32, 72, "zeros"), (1, 16, 2, 8, 96, 24, "reflect")]


@pytest.mark.parametrize("case", K333_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_k333_forward_and_weight_gradient_on_nc8(case):
    """3x3x3 stride-1 pad-1 layers: forward as (sample, frame) images with (time tap, channel) chunks, weight gradient as one
    transposed-read launch per time tap; reflect (all three dimensions) and zeros."""
    N, Cin, T, H, W, Cout, mode = case
    x, w = _bf(rnd(61, N, Cin, T, H, W)), _bf(rnd(62, Cout, Cin, 3, 3, 3, scale=(1.0 / (Cin * 27)) ** 0.5))
    b = rnd(63, Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    xp = F.pad(xr, (1,) * 6, mode="reflect") if mode == "reflect" else F.pad(xr, (1,) * 6)
    yr = F.conv3d(xp, wr, br)
    go = _bf(rnd(64, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    pl = ops._plan(xg.bfloat16(), wg, (1, 1, 1), (1, 1, 1), mode == "reflect")
    assert pl.k333_nc8 == (Cout > 4) and pl.k333_wgrad_nc8 == (Cout >= 64 and Cin >= 16), "routing of the 3x3x3 layer"
    assert pl.k333_dgrad_nc8 == (Cout != 24), "data gradient of the 3x3x3 layer on the pair-table form of the NC8 kernel"
    y = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode)
    (y * go.to(DEV)).sum().backward()
    assert y.dtype == torch.bfloat16 and y.shape == yr.shape
    rel_close(y.float(), yr, 4e-3, "3x3x3 NC8 forward")
    rel_close(wg.grad, wr.grad, 1e-4, "3x3x3 weight gradient")
    rel_close(bg.grad, br.grad, 1e-4, "bias gradient")
    rel_close(xg.grad, xr.grad, 5e-5, "data gradient")


def test_k333_final_fuse_shape_with_gradient_free_tail():
    """final_fuse (motion_autoencoder.py:131-141): 34 -> 32 channels, the last two input channels (the rastered sparse motion) carry
    no gradient (dgrad_channels = 32): three 16-channel chunks with 14 padded channels, data gradient for the leading 32 rows only."""
    N, Cin, T, H, W, Cout = 1, 34, 3, 16, 64, 32
    x, w = _bf(rnd(71, N, Cin, T, H, W)), _bf(rnd(72, Cout, Cin, 3, 3, 3, scale=(1.0 / (Cin * 27)) ** 0.5))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(F.pad(xr, (1,) * 6, mode="reflect"), wr)
    go = _bf(rnd(74, *yr.shape))
    (yr * go).sum().backward()
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    pl = ops._plan(xg.bfloat16(), wg, (1, 1, 1), (1, 1, 1), True, 32)
    assert pl.k333_nc8 and pl.k333_dgrad_nc8 and pl.dM == 32 and not pl.k333_wgrad_nc8
    y = ops.conv(xg, wg, None, stride=1, padding=1, padding_mode="reflect", dgrad_channels=32)
    (y * go.to(DEV)).sum().backward()
    rel_close(y.float(), yr, 4e-3, "final_fuse forward")
    rel_close(xg.grad[:, :32], xr.grad[:, :32], 5e-5, "data gradient of the leading 32 channels")
    assert float(xg.grad[:, 32:].abs().max()) == 0.0
    rel_close(wg.grad, wr.grad, 1e-4, "weight gradient")


@pytest.mark.parametrize("kind", ["bn", "in", "spade"])
def test_norm_kernels_write_the_nc8_side_output(kind):
    """The norm apply / backward-apply passes of the bf16 data path also write their result in the NC8 layout (attribute on the
    tensor, picked up by ops._to_nc8): bit-identical to the layout pass over the NCHW result, forward and backward."""
    N, C, H, W = 3, 40, 16, 32
    prev_on, ops._NC8_NORM = ops._NC8_NORM, True          # (off by default: see ops._NC8_NORM)
    try:
        _norm_side_output_case(kind, N, C, H, W)
    finally:
        ops._NC8_NORM = prev_on


def _norm_side_output_case(kind, N, C, H, W):
    x = rnd(81, N, C, H, W).to(DEV).bfloat16().requires_grad_(True)
    gamma, beta = rnd(82, C).to(DEV).requires_grad_(True), rnd(83, C).to(DEV).requires_grad_(True)
    gb = (rnd(84, N, 2 * C, H, W) * 0.3).to(DEV).bfloat16().requires_grad_(True)
    if kind == "bn":
        y = ops.batch_norm_act(x, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), act="lrelu")
    elif kind == "in":
        y = ops.instance_norm_act(x, gamma, beta, act="relu")
    else:
        y = ops.spade_norm_act(x, gb, act="lrelu")
    side = getattr(y, "_c2m_nc8", None)
    assert side is not None and side[0] == y._version
    ref = y.detach().view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(side[1], ref), "forward side output"
    assert ops._to_nc8(y) is side[1], "the conv picks the side output up instead of converting"
    # the same values as the plain apply kernel
    try:
        ops._NC8_NORM = False
        x2 = x.detach().clone().requires_grad_(True)
        if kind == "bn":
            y2 = ops.batch_norm_act(x2, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), act="lrelu")
        elif kind == "in":
            y2 = ops.instance_norm_act(x2, gamma, beta, act="relu")
        else:
            y2 = ops.spade_norm_act(x2, gb.detach(), act="lrelu")
    finally:
        ops._NC8_NORM = True
    assert getattr(y2, "_c2m_nc8", None) is None
    rel_close(y.float(), y2.float(), 8e-3, "NC8-writing apply vs the plain apply kernel (bf16 rounding of the same fp32 value)")
    # backward: a consumer that records what it is handed
    seen = {}

    class Spy(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            seen["g"] = g
            return g

    class Probe(torch.autograd.Function):          # stands where the convolution in front of the norm would stand
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            seen["dx"], seen["side"] = g, getattr(g, "_c2m_nc8", None)
            return g

    x3 = x.detach().clone().requires_grad_(True)
    h = Probe.apply(x3)
    if kind == "bn":
        y3 = ops.batch_norm_act(h, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), act="lrelu")
    elif kind == "in":
        y3 = ops.instance_norm_act(h, gamma, beta, act="relu")
    else:
        y3 = ops.spade_norm_act(h, gb, act="lrelu")
    y3.backward(rnd(85, N, C, H, W).to(DEV).bfloat16())
    assert seen["side"] is not None and seen["side"][0] == seen["dx"]._version
    dref = seen["dx"].view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(seen["side"][1], dref), "backward side output = NC8 form of dx"


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(2, 16, 8, 16), (3, 21, 4, 8), (1, 64, 2, 6, 8)])
def test_grad_to_nc8_matches_the_nchw_kernel_plus_layout_pass(shape, mode):
    """c2m_grad_to_nc8 (the convolution's own activation backward, mode 0; the VGG tap backward, mode 1) writes bit for bit what
    c2m_act_bwd / c2m_relu_tap_bwd followed by c2m_nchw_to_nc8 write."""
    from c2m_amd import _lib
    L = _lib.lib()
    y = rnd(101, *shape).to(DEV).bfloat16()
    gy = rnd(102, *shape).to(DEV).bfloat16()
    t = rnd(103, *shape).to(DEV).bfloat16()
    gl = torch.tensor([0.37], device=DEV)
    N, C = shape[0], shape[1]
    S = y.numel() // (N * C)
    gn = torch.empty((N, (C + 7) // 8) + tuple(shape[2:]) + (8,), device=DEV, dtype=torch.bfloat16)
    ref = torch.empty_like(y)
    if mode == 0:
        _lib.check(L.c2m_act_bwd(ops._p(y), ops._p(gy), ops._p(ref), y.numel(), ops.ACT["lrelu"], 0.2, 1, ops._stream()), "act_bwd")
        _lib.check(L.c2m_grad_to_nc8(0, ops._p(y), None, ops._p(gy), None, ops._p(gn), N, C, S, 0, ops.ACT["lrelu"], 0.2, ops._stream()), "g")
    else:
        _lib.check(L.c2m_relu_tap_bwd(ops._p(y), ops._p(t), ops._p(gy), ops._p(gl), ops._p(ref), y.numel(), 1, ops._stream()), "tap")
        _lib.check(L.c2m_grad_to_nc8(1, ops._p(y), ops._p(t), ops._p(gy), ops._p(gl), ops._p(gn), N, C, S, y.numel(), 0, 0.0, ops._stream()), "g")
    assert torch.equal(gn, ops._to_nc8(ref))


@pytest.mark.parametrize("case", [(2, 32, 16, 32, 64, 1, "reflect"), (2, 64, 16, 32, 64, 2, "reflect"), (1, 32, 24, 64, 64, 1, "zeros")])
def test_activation_backward_in_nc8_only_is_bit_identical_and_never_reads_its_nchw_storage(case):
    """A bf16 convolution with a fused activation whose backward runs on NC8 kernels only: the masked gradient exists in NC8 form
    alone (ops._virtual_grad).  Same data / weight / bias gradients, bit for bit, as with the NCHW act_bwd + layout pass -- also
    with the unused NCHW storage poisoned with NaN."""
    N, Cin, H, W, Cout, stride, mode = case
    k = 3 if stride == 1 else 4
    x = _bf(rnd(111, N, Cin, H, W)).to(DEV).bfloat16()
    w = _bf(rnd(112, Cout, Cin, k, k, scale=(1.0 / (Cin * k * k)) ** 0.5)).to(DEV)
    b = rnd(113, Cout, scale=0.1).to(DEV)
    outs = []
    for grad_nc8, poison in ((False, False), (True, False), (True, True)):
        ops._NC8_GRAD, ops._NC8_POISON = grad_nc8, poison
        ops._geom_cache.clear()
        try:
            xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            pl = ops._plan(xg, wg, (1, stride, stride), (0, 1, 1), mode == "reflect")
            elig = ops._bwd_reads_only_nc8(pl, True, True)
            assert elig == grad_nc8 or not grad_nc8, "the case must be eligible when the knob is on"
            if grad_nc8:
                assert elig
            y = ops.conv(xg, wg, bg, stride=stride, padding=1, padding_mode=mode, act="lrelu")
            y.backward(_bf(rnd(114, *y.shape)).to(DEV).bfloat16())
            outs.append((xg.grad.clone(), wg.grad.clone(), bg.grad.clone()))
        finally:
            ops._NC8_GRAD, ops._NC8_POISON = True, False
            ops._geom_cache.clear()
    for o in outs[1:]:
        for a, b_ in zip(outs[0], o):
            assert torch.isfinite(b_).all() and torch.equal(a, b_)


@pytest.mark.parametrize("spade", [False, True])
def test_residual_blocks_feed_their_convolutions_in_nc8_only(spade):
    """ResidualBlock / ResidualSpadeBlock hand their norm + activation results to the convolution behind them as NC8-only tensors
    (ops.conv_consumer): output and every gradient bit-identical to the NCHW hand-over, also with the unwritten NCHW storage
    poisoned with NaN."""
    from c2m_amd.modules.layers.residual_block import ResidualBlock, ResidualSpadeBlock
    x = _bf(rnd(121, 2, 64, 16, 32)).to(DEV).bfloat16()
    cond = _bf(rnd(122, 2, 16, 16, 32)).to(DEV).bfloat16()
    outs = []
    for grad_nc8, poison in ((False, False), (True, False), (True, True)):
        ops._NC8_GRAD, ops._NC8_POISON = grad_nc8, poison
        ops._geom_cache.clear()
        try:
            torch.manual_seed(3)
            blk = (ResidualSpadeBlock(16, 64, 64, 3, 1, {}) if spade else ResidualBlock(64, 64, 3, 1)).to(DEV).train()
            xg = x.clone().requires_grad_(True)
            if grad_nc8:
                w1 = blk.conv1.weight
                pl = ops._plan(xg, w1, (1, 1, 1), (0, 1, 1), True)
                assert ops._fwd_reads_only_nc8(pl, True), "the block's 3x3 layers must be eligible for an NC8-only input"
            y = blk(xg, cond) if spade else blk(xg)
            y.backward(_bf(rnd(123, *y.shape)).to(DEV).to(y.dtype))
            outs.append([y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None])
        finally:
            ops._NC8_GRAD, ops._NC8_POISON = True, False
            ops._geom_cache.clear()
    for o in outs[1:]:
        assert len(o) == len(outs[0])
        for a, b_ in zip(outs[0], o):
            assert torch.isfinite(b_).all() and torch.equal(a, b_)


BLOCK_CASES = ["down2d_s2", "down2d_s1", "same2d", "same3d", "up2d"]


@pytest.mark.parametrize("kind", BLOCK_CASES)
def test_blocks_hand_the_norm_gradient_back_in_nc8_only(kind):
    """conv -> norm -> activation blocks (DownBlock2d, SameBlock2d / 3d, UpBlock2d): the convolution's output is a temporary of the
    block, so the norm hands its dx to the convolution's backward in NC8 form alone (private_input) where that backward runs on NC8
    kernels.  Output and every gradient bit-identical to the NCHW hand-over, also with the unwritten NCHW storage NaN-poisoned."""
    from c2m_amd.modules.layers.down_block import DownBlock2d
    from c2m_amd.modules.layers.same_block import SameBlock2d, SameBlock3d
    from c2m_amd.modules.layers.up_block import UpBlock2d
    make = {
        "down2d_s2": (lambda: DownBlock2d(32, 64, kernel_size=(4, 4), stride=(2, 2), padding=1, padding_mode="reflect"), (2, 32, 32, 64)),
        "down2d_s1": (lambda: DownBlock2d(32, 64, padding_mode="reflect"), (2, 32, 16, 32)),
        "same2d": (lambda: SameBlock2d(64, 64, padding_mode="reflect"), (2, 64, 16, 32)),
        "same3d": (lambda: SameBlock3d(32, 64, kernel_size=3, padding=1, padding_mode="reflect"), (1, 32, 3, 16, 32)),
        "up2d": (lambda: UpBlock2d(32, 64, padding_mode="reflect", reshape_3d=False, input_2d=True), (2, 32, 8, 16)),
    }[kind]
    x = _bf(rnd(131, *make[1])).to(DEV).bfloat16()
    outs, used = [], []
    for grad_nc8, poison in ((False, False), (True, False), (True, True)):
        ops._NC8_GRAD, ops._NC8_POISON = grad_nc8, poison
        ops._geom_cache.clear()
        try:
            torch.manual_seed(5)
            blk = make[0]().to(DEV).train()
            xg = x.clone().requires_grad_(True)
            y = blk(xg)
            used.append(bool(getattr(y.grad_fn, "dx_nc8_only", False)))
            y.backward(_bf(rnd(132, *y.shape)).to(DEV).to(y.dtype))
            outs.append([y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None])
        finally:
            ops._NC8_GRAD, ops._NC8_POISON = True, False
            ops._geom_cache.clear()
    assert used == [False, True, True], f"the block's norm must take the NC8-only hand-over when the knob is on: {used}"
    for o in outs[1:]:
        assert len(o) == len(outs[0])
        for a, b_ in zip(outs[0], o):
            assert torch.isfinite(b_).all() and torch.equal(a, b_)


@pytest.mark.parametrize("shape", [(2, 16, 4, 8), (1, 21, 6, 12), (3, 64, 8, 16)])
def test_upsample2x_nc8_matches_the_nchw_kernel_plus_layout_pass(shape):
    from c2m_amd import _lib
    L = _lib.lib()
    N, C, H, W = shape
    x = rnd(141, *shape).to(DEV).bfloat16()
    y = torch.empty(N, C, 2 * H, 2 * W, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.c2m_upsample2x_fwd(ops._p(x), ops._p(y), N * C, H, W, 1, ops._stream()), "up")
    CB = (C + 7) // 8
    yn = torch.empty(N, CB, 2 * H, 2 * W, 8, device=DEV, dtype=torch.bfloat16)
    _lib.check(L.c2m_upsample2x_nc8(ops._p(ops._to_nc8(x)), ops._p(yn), N * CB, H, W, ops._stream()), "up nc8")
    assert torch.equal(yn, ops._to_nc8(y))
