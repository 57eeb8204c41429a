"""GPU parity tests (run with -m gpu on the MI355X box): every HIP op, through the C ABI, against the oracle /
golden vectors.  Index/mask paths bit-exact; fp32 elementwise rtol 1e-5; conv / norm reductions 1e-4 (stated per test)."""
import ctypes

import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from c2m_amd import ops
from oracle import c2m_oracle as O
from oracle import build as oracle_build
from golden_io import Case, names
from gpu_util import close, rel_close, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(t):
    return t.to(DEV)


# ----------------------------------------------------------------------------------------------- convolution
CONV_CASES = [
    # (x shape, Cout, kernel, stride, pad, mode)
    ((2, 6, 12, 16), 8, (4, 4), 2, 1, "reflect"),
    ((2, 6, 12, 16), 8, (4, 4), 2, 1, "zeros"),        # stride-2 dgrad: 4 parity classes batched into one launch
    ((2, 3, 10, 12), 4, (7, 7), 1, 3, "reflect"),
    ((1, 32, 9, 11), 3, (7, 7), 1, 3, "zeros"),
    ((2, 5, 9, 11), 7, (3, 3), 1, 1, "reflect"),
    ((2, 16, 16, 32), 32, (3, 3), 1, 1, "zeros"),
    ((3, 48, 8, 16), 80, (3, 3), 1, 1, "reflect"),
    ((2, 64, 8, 8), 130, (3, 3), 1, 1, "reflect"),
    ((2, 8, 6, 8), 4, (1, 1), 1, 0, "zeros"),
    ((10, 32, 16, 32), 16, (1, 1), 1, 0, "zeros"),     # 1x1, one K-step: packed dgrad weights are a pure transpose
    ((8, 512, 2, 4), 512, (3, 3), 1, 1, "reflect"),    # 64 pixels, K = 4608: split-K path (fwd and dgrad)
    ((2, 64, 4, 8), 96, (4, 4), 2, 1, "reflect"),      # stride-2 dgrad classes sharing one split-K slab set
    ((2, 8, 96, 96), 3, (3, 3), 1, 1, "reflect"),       # thin output (<= 4 channels, many pixels): vector-ALU kernels
    ((1, 32, 128, 160), 1, (3, 3), 1, 1, "reflect"),
    ((1, 6, 128, 144), 3, (7, 7), 1, 3, "zeros"),
    ((1, 3, 128, 160), 16, (3, 3), 1, 1, "zeros"),      # thin DATA gradient (3 input channels): row-blocked kernel, dx descending
    ((2, 5, 64, 132), 2, (7, 7), 1, 3, "reflect"),      # row-blocked thin kernel with reflected rows / columns
    ((1, 9, 128, 132), 1, (3, 3), 1, 1, "zeros"),       # row-blocked thin wgrad, channel count not a multiple of its block
    ((2, 10, 64, 136), 4, (3, 3), 1, 1, "reflect"),
    ((1, 5, 3, 40, 160), 2, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),
    ((2, 32, 20, 64), 48, (3, 3), 1, 1, "reflect"),      # LDS-patch kernel, 64-row tile, reflect patch loads
    ((1, 48, 12, 96), 130, (3, 3), 1, 1, "zeros"),      # LDS-patch kernel, 128-row tiles, 3 channel chunks
    ((1, 16, 128, 256), 16, (3, 3), 1, 1, "reflect"),   # patch dgrad over the padded 130x258 domain (partial tiles)
    ((2, 160, 16, 32), 96, (3, 3), 1, 1, "zeros"),      # LDS-patch kernel with split-K over channel chunks (fwd + dgrad)
    ((2, 21, 16, 32), 64, (4, 4), 2, 1, "reflect"),
    ((1, 32, 16, 32), 1, (3, 3), 1, 1, "reflect"),
    ((5, 40, 4, 8), 40, (3, 3), 1, 1, "reflect"),
    ((2, 5, 7, 9), 6, (3, 3), 2, 1, "zeros"),          # odd extents, stride 2 with k < 2*s classes
    ((1, 5, 5, 12, 16), 6, (4, 4, 4), (2, 2, 2), (1, 1, 1), "reflect"),
    ((2, 2, 5, 8, 12), 4, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),
    ((2, 4, 1, 4, 8), 6, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect"),
    ((2, 4, 1, 2, 4), 4, (1, 3, 3), (1, 1, 1), (0, 1, 1), "reflect"),
    ((2, 6, 5, 6, 8), 4, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),
    ((1, 34, 5, 16, 32), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),
    ((1, 40, 24, 64), 200, (3, 3), 1, 1, "reflect"),    # bf16 LDS-patch kernel: 2 row blocks of 128, padded last chunk
    # bf16 stride-2 weight gradient on the 16-byte-load kernel (every second element of a 16-element run; Wo % 8 == 0):
    # several groups per row, one group per row (left AND right end), zeros and reflect, 64- / 128- / 32-row tiles, 3-D
    # (4,4,4) stride (2,2,2) reflect with an EVEN frame count and a one-split data gradient: all eight parity classes have the same
    # extents and run as ONE class-batched two-target launch.  Class 6's (po_t, po_y) used to sit on geom[90] / geom[91], which every
    # call overwrites with the element-type flags: a quarter of the odd columns of dX was wrong (found by tools/fuzz_conv.py, seed 23)
    ((3, 16, 2, 6, 96), 8, (4, 4, 4), (2, 2, 2), (1, 1, 1), "reflect"),
    ((2, 24, 16, 64), 72, (4, 4), 2, 1, "reflect"),
    ((3, 8, 12, 16), 136, (4, 4), 2, 1, "zeros"),
    ((2, 16, 3, 8, 32), 24, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),
]


def _ref_conv(x, w, b, stride, pad, mode, act):
    nd = x.dim() - 2
    pads = (pad,) * nd if isinstance(pad, int) else tuple(pad)
    if mode == "reflect" and any(pads):
        tup = []
        for p in reversed(pads):
            tup += [p, p]
        x = F.pad(x, tuple(tup), mode="reflect")
        pads = (0,) * nd
    y = (F.conv2d if nd == 2 else F.conv3d)(x, w, b, stride=stride, padding=pads)
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2)
    elif act == "relu":
        y = F.relu(y)
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    return y


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"x{c[0]}_co{c[1]}_k{c[2]}_s{c[3]}_{c[5]}")
@pytest.mark.parametrize("act", [None, "lrelu"])
def test_conv_fwd_bwd(case, act):
    xs, cout, k, stride, pad, mode = case
    seed = zlib.crc32(str(case).encode()) % 10000            # (hash() of a str changes from process to process)
    x = rnd(seed, *xs)
    w = rnd(seed + 1, cout, xs[1], *k, scale=(1.0 / (xs[1] * int(np.prod(k))) ** 0.5))
    b = rnd(seed + 2, cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    xg, wg, bg = (g(t).requires_grad_(True) for t in (x, w, b))
    y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode, act=act)
    ypre = _ref_conv(xr, wr, br, stride, pad, mode, None)
    go = rnd(seed + 3, *ypre.shape)
    if act == "lrelu":
        # LeakyReLU's derivative is a step: where the reference pre-activation is within rounding of 0 the product's own
        # sign decides (either is a valid fp32 answer); everywhere else the reference's sign is required
        amb = ypre.detach().abs() < 1e-5 * float(ypre.detach().abs().max())
        pos = torch.where(amb, y.detach().cpu() > 0, ypre.detach() > 0)
        slope = torch.where(pos, torch.ones_like(go), torch.full_like(go, 0.2))
        yr = F.leaky_relu(ypre, 0.2)
        (ypre * (go * slope)).sum().backward()
    else:
        yr = ypre
        (yr * go).sum().backward()
    (y * g(go)).sum().backward()
    torch.cuda.synchronize()
    rel_close(y, yr, 2e-5, "conv fwd")                 # fp32 MFMA = fmaf chain; only the summation order differs
    rel_close(xg.grad, xr.grad, 5e-5, "conv dgrad")
    rel_close(wg.grad, wr.grad, 1e-4, "conv wgrad")
    # a bias gradient is a plain sum over N*H*W values of either sign: its rounding error scales with sum |go|, not with
    # the (cancelled) result
    rel_close(bg.grad, br.grad, 1e-4, "conv bias grad", floor=1e-3 * float(go.abs().sum()) / cout)


WINO_CASES = [c for c in CONV_CASES if len(c[0]) == 4 and c[2] == (3, 3) and c[3] == 1 and c[4] == 1] + [
    c for c in CONV_CASES if len(c[0]) == 5 and c[2] == (3, 3, 3) and c[3] == (1, 1, 1) and c[4] == (1, 1, 1)] + [
    ((2, 12, 3, 16, 32), 40, (3, 3, 3), (1, 1, 1), (1, 1, 1), "zeros"),     # 3x3x3 as 2-D Winograd over (time tap, channel)
    ((1, 34, 5, 16, 32), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "zeros"),
    ((2, 12, 1, 16, 32), 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), "zeros"),     # ONE frame: the time taps -1 / +1 only meet zeros
    ((3, 40, 24, 48), 70, (3, 3), 1, 1, "reflect"),      # channels not multiples of 8 / 64: zero-padded U, partial M tile
    ((2, 16, 10, 20), 8, (3, 3), 1, 1, "zeros"),         # partial 8x16 regions on both axes
    ((3, 40, 6, 48), 70, (3, 3), 1, 1, "zeros"),         # Winograd wgrad: partial channel tiles, 27 regions over 2 splits
    ((2, 130, 16, 32), 136, (3, 3), 1, 1, "reflect"),    # Winograd wgrad: every region touches the border, 3 x 3 channel tiles
    ((2, 64, 32, 64), 96, (3, 3), 1, 1, "reflect"),      # Winograd wgrad: several chunks per workgroup, reflected patches
    # region shapes other than 8 x 16 (c2m_wino_regions): 18 x 34 padded domain as 3 x 10 tiles on the 32-row kernel
    # (data gradient of 32 input channels), an 18 x 34 forward domain with <= 32 output channels, 34 x 66 as 6 x 5 tiles
    ((2, 32, 16, 32), 64, (3, 3), 1, 1, "reflect"),
    ((2, 40, 18, 34), 32, (3, 3), 1, 1, "zeros"),
    ((1, 64, 32, 64), 64, (3, 3), 1, 1, "reflect"),
    ((2, 24, 3, 8, 16), 40, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),    # 3x3x3 data gradient over 10 x 18 frames: 2 regions of 3 x 9 tiles;
                                                                             # T = 3: the middle frame sums 5 (frame, tap) pairs
    ((1, 16, 2, 16, 32), 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),    # T = 2: both frames are mirror targets (3 pairs each)
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: f"x{c[0]}_co{c[1]}_{c[5]}")
def test_conv_winograd_forced(case, monkeypatch):
    """Winograd F(2x2,3x3) kernel on EVERY 3x3 stride-1 pad-1 shape of the suite, whatever the auto heuristic would
    pick: forward, zero-pad data gradient, reflect data gradient over the padded domain (two-target epilogue)."""
    monkeypatch.setattr(ops, "_WINO", "force")
    monkeypatch.setattr(ops, "_WINO_WGRAD", "force")
    monkeypatch.setattr(ops, "_RING", "off")         # (the interior + ring route: test_conv_reflect_ring_dgrad)
    ops._geom_cache.clear()
    try:
        xs, cout, k, stride, pad, mode = case
        seed = zlib.crc32(("wino" + str(case)).encode()) % 10000
        x = rnd(seed, *xs)
        w = rnd(seed + 1, cout, xs[1], *k, scale=(1.0 / (xs[1] * 9) ** 0.5))
        b = rnd(seed + 2, cout, scale=0.1)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        yr = _ref_conv(xr, wr, br, stride, pad, mode, None)
        go = rnd(seed + 3, *yr.shape)
        (yr * go).sum().backward()
        xg, wg, bg = (g(t).requires_grad_(True) for t in (x, w, b))
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        is3d = len(xs) == 5      # 3x3x3: 2-D Winograd kernels over (time tap, channel) virtual channels, image = (n, t)
        pl = ops._plan(xg, wg, (1, 1, 1), (1, 1, 1) if is3d else (0, 1, 1), mode == "reflect")
        assert pl.wino_fwd and pl.wino_dgrad
        keff = (3 if is3d else 1) * xs[1]            # input channels of the Winograd weight gradient (virtual ones in 3-D)
        assert pl.wino_wgrad == (xs[-2] % 2 == 0 and xs[-1] % 16 == 0 and (keff * cout) % 4 == 0)
        assert not is3d or (pl.wino3d and pl.wino_wgrad3d == pl.wino_wgrad)
        (y * g(go)).sum().backward()
        rel_close(y, yr, 2e-5, "winograd fwd")
        rel_close(xg.grad, xr.grad, 5e-5, "winograd dgrad")
        rel_close(wg.grad, wr.grad, 1e-4, "wgrad (Winograd where H % 2 == 0 and W % 16 == 0, else the direct kernel)")
        rel_close(bg.grad, br.grad, 1e-4, "bias grad", floor=1e-3 * float(go.abs().sum()) / cout)
    finally:
        ops._geom_cache.clear()


WINO4_CASES = [c for c in WINO_CASES if len(c[0]) == 4] + [
    ((2, 256, 16, 32), 256, (3, 3), 1, 1, "reflect"),    # the 16 x 32 bottleneck blocks: one region per image, 32 chunks
    ((1, 512, 16, 32), 128, (3, 3), 1, 1, "zeros"),      # VGG conv4-like depth: 64 chunks
    ((1, 72, 40, 72), 80, (3, 3), 1, 1, "zeros"),        # partial regions on both axes (3 x 3 regions of 16 x 32), partial M tile
    ((1, 17, 16, 32), 64, (3, 3), 1, 1, "reflect"),      # K not a multiple of 8: the last chunk re-reads the last channel against zero U rows
    ((3, 8, 16, 32), 48, (3, 3), 1, 1, "zeros"),         # ONE chunk (prologue + one interval)
]


@pytest.mark.parametrize("case", WINO4_CASES, ids=lambda c: f"x{c[0]}_co{c[1]}_{c[5]}")
def test_conv_winograd_f4x4_forced(case, monkeypatch):
    """Winograd F(4x4,3x3) kernel (conv_wino4.hip) on every 2-D 3x3 stride-1 pad-1 shape of the suite: forward, zero-pad data
    gradient, reflect data gradient over the padded domain (two-target stores), at the SAME gates as the other conv kernels
    (measured on the GPU: 2e-6 ... 4e-6 of the tensor scale, tools/bench_wino4.py)."""
    monkeypatch.setattr(ops, "_WINO", "force")
    monkeypatch.setattr(ops, "_WINO4", "force")
    monkeypatch.setattr(ops, "_RING", "off")
    ops._geom_cache.clear()
    try:
        xs, cout, k, stride, pad, mode = case
        seed = zlib.crc32(("wino4" + str(case)).encode()) % 10000
        x = rnd(seed, *xs)
        w = rnd(seed + 1, cout, xs[1], *k, scale=(1.0 / (xs[1] * 9) ** 0.5))
        b = rnd(seed + 2, cout, scale=0.1)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        yr = _ref_conv(xr, wr, br, stride, pad, mode, None)
        go = rnd(seed + 3, *yr.shape)
        (yr * go).sum().backward()
        xg, wg, bg = (g(t).requires_grad_(True) for t in (x, w, b))
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        pl = ops._plan(xg, wg, (1, 1, 1), (0, 1, 1), mode == "reflect")
        assert pl.wino_fwd and pl.wino_dgrad and pl.wino4_fwd and pl.wino4_dgrad
        (y * g(go)).sum().backward()
        rel_close(y, yr, 2e-5, "F(4x4,3x3) fwd")
        rel_close(xg.grad, xr.grad, 5e-5, "F(4x4,3x3) dgrad")
        rel_close(wg.grad, wr.grad, 1e-4, "wgrad next to the F(4x4,3x3) launches")
        # epilogue activation (applied after the bias, inside the inverse-transform pass); elements whose pre-activation is within
        # rounding of 0 may take either slope
        with torch.no_grad():
            ya = ops.conv(g(x), g(w), g(b), stride=stride, padding=pad, padding_mode=mode, act="lrelu")
        rel_close(ya, F.leaky_relu(yr.detach(), 0.2), 2e-5, "F(4x4,3x3) fwd + lrelu")
        # twice the same launch: bit-identical (fixed summation order, no atomics)
        with torch.no_grad():
            yb = ops.conv(g(x), g(w), g(b), stride=stride, padding=pad, padding_mode=mode, act="lrelu")
        assert torch.equal(ya, yb)
    finally:
        ops._geom_cache.clear()


def _bf(t):
    return t.bfloat16().float()


RING_CASES = [c for c in WINO4_CASES if c[5] == "reflect" and min(c[0][2:]) >= 4] + [
    ((3, 33, 4, 4), 50, (3, 3), 1, 1, "reflect"),        # smallest map: rows 1 / H-2 adjacent, every row target but x = 0, 3 is a corner
    ((2, 40, 6, 10), 24, (3, 3), 1, 1, "reflect"),       # 24 reduction channels: the second chunk half empty; 40 rows: partial 64-row tile
    ((5, 96, 32, 64), 128, (3, 3), 1, 1, "reflect"),     # line tiles that span images (5 x 64 = 2.5 pixel tiles), 8 chunks
    ((2, 64, 64, 128), 64, (3, 3), 1, 1, "reflect"),     # the 64 x 128 maps the auto rule moves to F(4x4) on the exact domain
    ((1, 130, 8, 130), 72, (3, 3), 1, 1, "reflect"),     # W > 128: a row spans two pixel tiles; M = 130: three 64-row tiles
]


@pytest.mark.parametrize("form", ["buffer", "inplace"])
@pytest.mark.parametrize("w4", ["F(2x2)", "F(4x4)"])
@pytest.mark.parametrize("case", RING_CASES, ids=lambda c: f"x{c[0]}_co{c[1]}")
def test_conv_reflect_ring_dgrad(case, w4, form, monkeypatch):
    """Reflect-pad data gradient as interior + ring (round 5): the Winograd kernels run the zero-padded "same" data gradient over
    the EXACT H x W domain and c2m_reflect_ring_dgrad (conv_ring.hip) adds what the pad ring of the padded gradient mirrors onto
    rows 1, H-2 / columns 1, W-2 -- against torch's reflection_pad2d + conv2d autograd on the CPU at the data-gradient gate, the
    weight gradient beside it, and twice the same bits (every element of dX has one writer)."""
    monkeypatch.setattr(ops, "_WINO", "force")
    monkeypatch.setattr(ops, "_WINO4", "force" if w4 == "F(4x4)" else "off")
    monkeypatch.setattr(ops, "_RING", "force")
    # "buffer" (product): ring terms written to a compact buffer, added by the Winograd epilogue; "inplace": the first form of the
    # round (ring launch read-modify-writes dX + corner part) -- same sums up to the order of three additions at the corner targets
    monkeypatch.setattr(ops, "_RING_BUFFER", form == "buffer")
    ops._geom_cache.clear()
    try:
        xs, cout, k, stride, pad, mode = case
        seed = zlib.crc32(("ring" + str(case)).encode()) % 10000
        x = rnd(seed, *xs)
        w = rnd(seed + 1, cout, xs[1], *k, scale=(1.0 / (xs[1] * 9) ** 0.5))
        xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
        yr = _ref_conv(xr, wr, None, stride, pad, mode, None)
        go = rnd(seed + 3, *yr.shape)
        (yr * go).sum().backward()
        grads = []
        for _ in range(2):
            xg, wg = (g(t).requires_grad_(True) for t in (x, w))
            y = ops.conv(xg, wg, None, stride=stride, padding=pad, padding_mode=mode)
            pl = ops._plan(xg, wg, (1, 1, 1), (0, 1, 1), True)
            assert pl.wino_dgrad and pl.ring_dgrad and pl.wino4_dgrad == (w4 == "F(4x4)")
            (y * g(go)).sum().backward()
            grads.append(xg.grad.clone())
        rel_close(grads[0], xr.grad, 5e-5, f"reflect dgrad: {w4} interior + ring")
        # the ring rows / columns alone (the interior dominates the tensor scale): rows 1, H-2 and columns 1, W-2
        H, W = xs[2], xs[3]
        for sl in ((slice(None), slice(None), [1, H - 2], slice(None)), (slice(None), slice(None), slice(None), [1, W - 2])):
            rel_close(grads[0][sl], xr.grad[sl], 5e-5, f"reflect dgrad, ring targets: {w4}")
        rel_close(wg.grad, wr.grad, 1e-4, "wgrad beside the ring route")
        assert torch.equal(grads[0], grads[1]), "interior + ring must be bit-repeatable"
    finally:
        ops._geom_cache.clear()


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"x{c[0]}_co{c[1]}_k{c[2]}_s{c[3]}_{c[5]}")
def test_conv_bf16_mode(case):
    """bf16 operand mode (BASELINE configs[2-4]): with inputs that are exactly representable in bf16 every product is
    exact in fp32, so the result must equal the fp32 convolution up to summation order -- this pins the bf16 LDS image,
    the K-slot permutation and the 32x32x16 operand layout; arbitrary inputs then differ only by the operand rounding."""
    xs, cout, k, stride, pad, mode = case
    seed = zlib.crc32(str(case).encode()) % 10000            # (hash() of a str changes from process to process)
    x, w = _bf(rnd(seed, *xs)), _bf(rnd(seed + 1, cout, xs[1], *k, scale=(1.0 / (xs[1] * int(np.prod(k))) ** 0.5)))
    b = rnd(seed + 2, cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = _ref_conv(xr, wr, br, stride, pad, mode, None)
    go = _bf(rnd(seed + 3, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (g(t).requires_grad_(True) for t in (x, w, b))
    with ops.conv_precision("bf16"):
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        (y * g(go)).sum().backward()
    # bf16 data path: the result tensor itself is bf16 (<= 4-channel heads stay fp32) -- one RNE rounding of the exact sum
    assert y.dtype == (torch.bfloat16 if cout > 4 else torch.float32)
    rel_close(y.float(), yr, 4e-3 if cout > 4 else 2e-5, "bf16-mode conv fwd (representable inputs)")
    assert xg.grad.dtype == torch.float32 and wg.grad.dtype == torch.float32      # gradients take their tensor's type
    rel_close(xg.grad, xr.grad, 5e-5, "bf16-mode dgrad")
    rel_close(wg.grad, wr.grad, 1e-4, "bf16-mode wgrad")
    rel_close(bg.grad, br.grad, 1e-4, "bf16-mode bias grad")
    # arbitrary fp32 inputs: operands rounded to 8 mantissa bits -> relative error of a dot product ~ 2^-8 / sqrt(K)
    x2, w2 = rnd(seed + 5, *xs), rnd(seed + 6, cout, xs[1], *k, scale=(1.0 / (xs[1] * int(np.prod(k))) ** 0.5))
    with ops.conv_precision("bf16"):
        y2 = ops.conv(g(x2), g(w2), None, stride=stride, padding=pad, padding_mode=mode)
    rel_close(y2.float(), _ref_conv(x2, w2, None, stride, pad, mode, None), 1e-2, "bf16-mode conv fwd (fp32 inputs)")
    if cout > 4:       # <= 4 output channels run on the fp32 vector-ALU kernel in either mode (more precise, not less)
        ref2 = _ref_conv(_bf(x2), _bf(w2), None, stride, pad, mode, None)
        rel_close(y2.float(), ref2, 4e-3, "bf16 rounding is RNE of both operands and of the result")
        # a bf16 INPUT tensor is consumed as it is: same bits as the cast the op applies to an fp32 input
        with ops.conv_precision("bf16"):
            y3 = ops.conv(g(x2).bfloat16(), g(w2), None, stride=stride, padding=pad, padding_mode=mode)
        assert torch.equal(y3, y2)


@pytest.mark.parametrize("cin,cout", [(40, 200), (200, 45)])
def test_conv_bf16_patch_never_reads_past_the_input(cin, cout):
    """the bf16 LDS-patch kernel carries the channel in the scalar offset of its buffer loads, which the
    hardware does not range-check.  With Cin (forward) or Cout (data gradient) not a multiple of 16 the padded channels of the
    last chunk must read 0, not whatever lies behind the tensor: the inputs here are views whose allocation continues with
    NaNs, and no output may become NaN."""
    H, W = 24, 64
    # the kernels gather bf16 tensors (the bf16 data path): the views below are consumed as they are, NaNs right behind them
    pool = torch.full((2 * cin * H * W + 4096,), float("nan"), device=DEV, dtype=torch.bfloat16)
    x = pool[:cin * H * W].view(1, cin, H, W)
    x.copy_(g(_bf(rnd(7, 1, cin, H, W))))
    w = g(_bf(rnd(8, cout, cin, 3, 3, scale=(1.0 / (cin * 9)) ** 0.5)))
    gpool = torch.full((2 * cout * H * W + 4096,), float("nan"), device=DEV, dtype=torch.bfloat16)
    go = gpool[:cout * H * W].view(1, cout, H, W)
    go.copy_(g(_bf(rnd(9, 1, cout, H, W))))
    xg, wg = x.requires_grad_(True), w.requires_grad_(True)
    with ops.conv_precision("bf16"):
        pl = ops._plan(xg, wg, (1, 1, 1), (0, 1, 1), False)
        assert pl.fwd_patch or any(c["patch"] for c in pl.classes), "the case must run on the bf16 patch kernel"
        y = ops.conv(xg, wg, None, stride=1, padding=1)
        y.backward(go)
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(xg.grad).all()) and bool(torch.isfinite(wg.grad).all())
    xr, wr = x.detach().float().cpu().requires_grad_(True), w.detach().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=1)
    yr.backward(go.float().cpu())
    assert y.dtype == torch.bfloat16 and xg.grad.dtype == torch.bfloat16
    rel_close(y.float(), yr, 4e-3, "bf16 patch fwd")
    rel_close(xg.grad.float(), xr.grad, 4e-3, "bf16 patch dgrad")


def test_conv_stride2_fuzz():
    """Random small stride-2 layers (the tiny e2e fixtures live here): batched parity classes, split-K, thin rows."""
    rs = np.random.RandomState(7)
    for it in range(40):
        three_d = it % 4 == 3
        cin, cout = int(rs.randint(1, 41)), int(rs.randint(1, 41))
        h, w_ = int(rs.randint(2, 11)) * 2, int(rs.randint(2, 11)) * 2
        n = int(rs.randint(1, 4))
        mode = "reflect" if it % 2 == 0 else "zeros"
        if three_d:
            t = int(rs.randint(2, 6))
            kt = int(rs.choice([3, 4]))
            xs, k, stride, pad = (n, cin, t, h, w_), (kt, 4, 4), ((2 if kt == 4 else 1), 2, 2), (1, 1, 1)
            if kt == 4 and t < 2:
                continue
        else:
            xs, k, stride, pad = (n, cin, h, w_), (4, 4), 2, 1
        x = rnd(1000 + it, *xs)
        w = rnd(2000 + it, cout, cin, *k, scale=(1.0 / (cin * int(np.prod(k))) ** 0.5))
        b = rnd(3000 + it, cout, scale=0.1)
        xr, wr, br = (t_.clone().requires_grad_(True) for t_ in (x, w, b))
        yr = _ref_conv(xr, wr, br, stride, pad, mode, None)
        go = rnd(4000 + it, *yr.shape)
        (yr * go).sum().backward()
        xg, wg, bg = (g(t_).requires_grad_(True) for t_ in (x, w, b))
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        (y * g(go)).sum().backward()
        what = f"case {it}: x{xs} cout {cout} k{k} s{stride} {mode}"
        rel_close(y, yr, 2e-5, what + " fwd")
        rel_close(xg.grad, xr.grad, 5e-5, what + " dgrad")
        rel_close(wg.grad, wr.grad, 1e-4, what + " wgrad")
        rel_close(bg.grad, br.grad, 1e-4, what + " bias grad")


def test_conv_frozen_weight_pack_cache():
    """Frozen weights (VGG) are packed once; an in-place update or a new tensor in the same allocation must miss."""
    x = g(rnd(1, 2, 16, 8, 32))
    w = g(rnd(2, 16, 16, 3, 3, scale=0.1))
    ops._frozen_pack_cache.clear()
    y1 = ops.conv(x, w, None, stride=1, padding=1)
    n1 = len(ops._frozen_pack_cache)
    y2 = ops.conv(x, w, None, stride=1, padding=1)
    assert n1 == 1 and len(ops._frozen_pack_cache) == 1 and torch.equal(y1, y2)
    w.mul_(2.0)                                   # _version bump -> repack
    close(ops.conv(x, w, None, stride=1, padding=1), 2.0 * y1, 1e-6, 1e-6, "in-place update")
    ptr = w.data_ptr()
    del w
    w2 = g(rnd(3, 16, 16, 3, 3, scale=0.1))       # usually lands in the freed block
    yr = F.conv2d(x, w2, padding=1)
    rel_close(ops.conv(x, w2, None, stride=1, padding=1), yr, 2e-5, f"new tensor (same ptr: {w2.data_ptr() == ptr})")


def test_conv_big_wgrad_splitk():
    """Many pixels, few channels: exercises the split-K slabs + fixed-order reduction."""
    x, w = rnd(1, 4, 8, 64, 128), rnd(2, 16, 8, 3, 3, scale=0.1)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr)
    go = rnd(3, *yr.shape)
    (yr * go).sum().backward()
    xg, wg = g(x).requires_grad_(True), g(w).requires_grad_(True)
    y = ops.conv(xg, wg, None, stride=1, padding=1, padding_mode="reflect")
    (y * g(go)).sum().backward()
    rel_close(y, yr, 2e-5)
    rel_close(wg.grad, wr.grad, 2e-4, "split-K wgrad")
    rel_close(xg.grad, xr.grad, 5e-5)
    # determinism: same launch twice -> bit-identical weight gradient (no float atomics)
    xg2, wg2 = g(x).requires_grad_(True), g(w).requires_grad_(True)
    (ops.conv(xg2, wg2, None, stride=1, padding=1, padding_mode="reflect") * g(go)).sum().backward()
    assert torch.equal(wg.grad, wg2.grad) and torch.equal(xg.grad, xg2.grad)


# ----------------------------------------------------------------------------------------------- norm + act
@pytest.mark.parametrize("shape", [(3, 5, 6, 8), (2, 4, 5, 6, 8), (4, 16, 2, 4), (2, 3, 100, 90), (6, 8),
                                   # >= 16 partials per channel: the wave-per-channel finalize kernels
                                   (20, 6, 16, 24), (70, 3, 4, 4), (2, 3, 300, 300)])
@pytest.mark.parametrize("act", [None, "lrelu", "relu"])
def test_batch_norm_act(shape, act):
    x = rnd(1, *shape) * 2 + 3.0          # non-zero mean: the statistics must be cancellation-safe
    C = shape[1]
    gam, bet = 1 + 0.1 * rnd(2, C), 0.1 * rnd(3, C)
    rm, rv = 0.1 * rnd(4, C), 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(5))
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gam, bet))
    rmr, rvr = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    yr = F.leaky_relu(yr, 0.2) if act == "lrelu" else (F.relu(yr) if act == "relu" else yr)
    go = rnd(6, *shape)
    (yr * go).sum().backward()
    xg, gg, bg = (g(t).requires_grad_(True) for t in (x, gam, bet))
    rmg, rvg = g(rm), g(rv)
    y = ops.batch_norm_act(xg, gg, bg, rmg, rvg, act)
    (y * g(go)).sum().backward()
    close(y, yr, 1e-4, 1e-5, "bn fwd")
    close(rmg, rmr, 1e-5, 1e-6, "running_mean")
    close(rvg, rvr, 1e-4, 1e-6, "running_var")
    close(xg.grad, xr.grad, 1e-3, 1e-5, "bn dx")
    close(gg.grad, gr.grad, 1e-4, 1e-4, "bn dgamma")
    close(bg.grad, br.grad, 1e-4, 1e-4, "bn dbeta")


@pytest.mark.parametrize("shape", [(2, 5, 9, 11), (3, 8, 2, 4), (1, 4, 130, 70), (20, 5, 9, 11), (70, 3, 4, 4)])
@pytest.mark.parametrize("affine", [True, False])
def test_instance_norm_act(shape, affine):
    x = rnd(1, *shape) + 1.5
    C = shape[1]
    gam, bet = (1 + 0.1 * rnd(2, C), 0.1 * rnd(3, C)) if affine else (None, None)
    xr = x.clone().requires_grad_(True)
    gr, br = (gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)) if affine else (None, None)
    yr = F.leaky_relu(F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.2)
    go = rnd(4, *shape)
    (yr * go).sum().backward()
    xg = g(x).requires_grad_(True)
    gg, bg = (g(gam).requires_grad_(True), g(bet).requires_grad_(True)) if affine else (None, None)
    y = ops.instance_norm_act(xg, gg, bg, "lrelu")
    (y * g(go)).sum().backward()
    close(y, yr, 1e-4, 1e-5, "in fwd")
    close(xg.grad, xr.grad, 1e-3, 1e-5, "in dx")
    if affine:
        close(gg.grad, gr.grad, 1e-4, 1e-4)
        close(bg.grad, br.grad, 1e-4, 1e-4)


def test_spade_norm_act():
    x, gb = rnd(1, 2, 5, 6, 8) + 0.5, 0.3 * rnd(2, 2, 10, 6, 8)
    xr, gbr = x.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    ga, be = gbr.chunk(2, 1)
    yr = F.leaky_relu(F.instance_norm(xr, None, None, None, None, True, 0.1, 1e-5) * (1 + ga) + be, 0.2)
    go = rnd(3, *yr.shape)
    (yr * go).sum().backward()
    xg, gbg = g(x).requires_grad_(True), g(gb).requires_grad_(True)
    y = ops.spade_norm_act(xg, gbg, "lrelu")
    (y * g(go)).sum().backward()
    close(y, yr, 1e-4, 1e-5, "spade fwd")
    close(xg.grad, xr.grad, 1e-3, 1e-5, "spade dx")
    close(gbg.grad, gbr.grad, 1e-4, 1e-5, "spade d(gamma,beta)")


@pytest.mark.parametrize("shape", [(3, 5, 32, 64), (2, 6, 64, 128), (2, 3, 128, 256), (2, 4, 36, 40), (5, 2, 16, 64)])
@pytest.mark.parametrize("kind", ["plain", "affine", "spade"])
def test_instance_norm_one_launch_kernels(shape, kind):
    """Round 5: instance-norm planes of 1024 ... 32768 elements run statistics + apply as ONE launch (norm_inst_fused_kernel) and,
    without affine-parameter gradients and up to 8192 elements, the backward as one launch too (norm_inst_bwd_fused_kernel).
    Against torch on the CPU, and against the three-launch path of the same library (c2m_norm_set_fused(0)): bit-identical for
    planes of one 8192-element chunk (same arithmetic, same summation order), within rounding above."""
    from c2m_amd import _lib
    N, C, H, W = shape
    x = rnd(11, *shape) * 1.7 + 0.8
    gam, bet = (1 + 0.1 * rnd(12, C), 0.1 * rnd(13, C)) if kind == "affine" else (None, None)
    gb = 0.3 * rnd(14, N, 2 * C, H, W) if kind == "spade" else None
    go = rnd(15, *shape)
    xr = x.clone().requires_grad_(True)
    gr, br = (gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)) if kind == "affine" else (None, None)
    gbr = gb.clone().requires_grad_(True) if gb is not None else None
    yn = F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if gbr is not None:
        ga_, be_ = gbr.chunk(2, 1)
        yn = yn * (1 + ga_) + be_
    yr = F.leaky_relu(yn, 0.2)
    (yr * go).sum().backward()

    def run():
        xg = g(x).requires_grad_(True)
        gg, bg = (g(gam).requires_grad_(True), g(bet).requires_grad_(True)) if kind == "affine" else (None, None)
        gbg = g(gb).requires_grad_(True) if gb is not None else None
        y = ops.spade_norm_act(xg, gbg, "lrelu") if kind == "spade" else ops.instance_norm_act(xg, gg, bg, "lrelu")
        (y * g(go)).sum().backward()
        return [y.detach(), xg.grad] + ([gg.grad, bg.grad] if kind == "affine" else []) + ([gbg.grad] if gbg is not None else [])
    fused = run()
    old = _lib.lib().c2m_norm_set_fused(0)
    try:
        three = run()
    finally:
        _lib.lib().c2m_norm_set_fused(old)
    close(fused[0], yr, 1e-4, 1e-5, "one-launch instance norm fwd")
    close(fused[1], xr.grad, 1e-3, 1e-5, "one-launch instance norm dx")
    if kind == "affine":
        close(fused[2], gr.grad, 1e-4, 1e-4)
        close(fused[3], br.grad, 1e-4, 1e-4)
    if kind == "spade":
        close(fused[2], gbr.grad, 1e-4, 1e-5, "spade d(gamma, beta)")
    for a, b in zip(fused, three):
        if H * W <= 8192 and kind != "spade":
            assert torch.equal(a, b), "one chunk per plane: the one-launch kernels must reproduce the three-launch path bit for bit"
        elif H * W <= 8192:
            # SPADE: the compiler contracts xhat * (1 + gamma) + beta into FMAs differently in the two kernels -- last-bit differences
            close(a, b.cpu(), 2e-6, 2e-6, "one-launch vs three-launch path (SPADE)")
        else:
            close(a, b.cpu(), 1e-4, 1e-5, "one-launch vs three-launch path")


@pytest.mark.parametrize("shape", [(40, 6, 4, 8), (40, 5, 16, 32), (8, 7, 5, 8, 16), (24, 9), (3, 4, 50, 50), (40, 3, 8, 16)])
@pytest.mark.parametrize("act", ["lrelu", "relu"])
def test_batch_norm_small_volume_one_launch(shape, act):
    """Round 5: batch norm of a small channel volume (N * S <= 8192 elements per channel -- the deep encoder levels; the larger
    shapes of the list stay on the three-launch path) runs as ONE launch per direction (norm_bn_small_fused_kernel / _bwd_): against torch on the CPU
    and against the three-launch path of the same library (different summation order: within rounding), running statistics too."""
    from c2m_amd import _lib
    x = rnd(21, *shape) * 1.5 + 2.0
    C = shape[1]
    gam, bet = 1 + 0.1 * rnd(22, C), 0.1 * rnd(23, C)
    rm, rv = 0.1 * rnd(24, C), 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(25))
    go = rnd(26, *shape)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gam, bet))
    rmr, rvr = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    yr = F.leaky_relu(yr, 0.2) if act == "lrelu" else F.relu(yr)
    (yr * go).sum().backward()

    def run():
        xg, gg, bg = (g(t).requires_grad_(True) for t in (x, gam, bet))
        rmg, rvg = g(rm), g(rv)
        y = ops.batch_norm_act(xg, gg, bg, rmg, rvg, act)
        (y * g(go)).sum().backward()
        return [y.detach(), xg.grad, gg.grad, bg.grad, rmg, rvg]
    fused = run()
    old = _lib.lib().c2m_norm_set_fused(0)
    try:
        three = run()
    finally:
        _lib.lib().c2m_norm_set_fused(old)
    for a, ref, tol, what in zip(fused, (yr, xr.grad, gr.grad, br.grad, rmr, rvr), (1e-4, 1e-3, 1e-4, 1e-4, 1e-5, 1e-4),
                                 ("fwd", "dx", "dgamma", "dbeta", "running_mean", "running_var")):
        close(a, ref, tol, 1e-4 if what in ("dgamma", "dbeta") else 1e-5, f"one-launch bn {what}")
    for a, b, what in zip(fused, three, ("fwd", "dx", "dgamma", "dbeta", "running_mean", "running_var")):
        close(a, b.cpu(), 1e-4, 1e-4 if what in ("dx", "dgamma", "dbeta") else 1e-5, f"one-launch vs three-launch bn {what}")


# ----------------------------------------------------------------------------------------------- warping / resampling
@pytest.mark.parametrize("name", names("op_resample"))
def test_flow_warp_golden(name):
    c = Case(name)
    i = c.group("in")
    img, flow = g(i["image"]).requires_grad_(True), g(i["flow"]).requires_grad_(True)
    y = ops.flow_warp(img, flow)
    assert torch.equal(y.detach().cpu(), c.group("out")["y"]), "flow_warp forward must be bit-exact vs the reference"
    if "gout" in i:
        (y * g(i["gout"])).sum().backward()
        gin = c.group("gin")
        close(img.grad, gin["image"], 1e-5, 1e-6, "d image")
        close(flow.grad, gin["flow"], 1e-4, 1e-5, "d flow")


@pytest.mark.parametrize("shape", [(40, 64, 8, 16), (5, 3, 128, 256), (8, 256, 16, 32), (2, 5, 9, 13)])
def test_flow_warp_backward_is_bit_repeatable(shape):
    """No float atomics: d(image) is gathered through the inverted tap list in a fixed (dest pixel, corner) order and
    d(flow) is summed in a fixed channel order, so two launches give identical bits -- also for a strongly converging
    flow (long tap lists) and with the occlusion factor."""
    N, C, H, W = shape
    img, go = g(rnd(11, *shape)), g(rnd(12, *shape))
    occ = g(torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(13)))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    converge = torch.stack([(W / 2 - xs) * 0.9, (H / 2 - ys) * 0.9], 0).unsqueeze(0).repeat(N, 1, 1, 1)
    for flow in (rnd(14, N, 2, H, W, scale=2.5), converge + rnd(15, N, 2, H, W, scale=0.3)):
        grads = []
        for _ in range(2):
            ig, fg = img.clone().requires_grad_(True), g(flow).requires_grad_(True)
            (ops.flow_warp(ig, fg, occ) * go).sum().backward()
            grads.append((ig.grad.clone(), fg.grad.clone()))
        assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
        # and the values are right (CPU restatement; its own scatter order differs only in rounding)
        ir, fr = img.cpu().clone().requires_grad_(True), flow.clone().requires_grad_(True)
        ((O.resample(ir, fr) * occ.cpu()) * go.cpu()).sum().backward()
        rel_close(grads[0][0], ir.grad, 2e-5, "d image (converging flow)" if flow is not None else "")
        rel_close(grads[0][1], fr.grad, 2e-4, "d flow")


def test_flow_warp_backward_diverged_flow_is_bounded_and_exact():
    """a diverged flow field sends whole regions onto border / corner pixels -- buckets of 1e4+ taps.  The
    per-thread insertion sort would be O(n^2) there (seconds: looks like a hang); buckets above 32 entries are sorted by a
    workgroup-wide bitonic network instead.  Same fixed (dest pixel, corner) order, so still bit-repeatable and equal to
    the CPU restatement up to its own summation order."""
    import time
    N, C, H, W = 2, 3, 128, 256
    img, go = g(rnd(21, N, C, H, W)), g(rnd(22, N, C, H, W))
    flow = rnd(23, N, 2, H, W, scale=3.0)
    flow[0] += 4000.0                        # every pixel of image 0 samples the bottom-right corner: ONE bucket of 32768 x 1 taps
    flow[1, 0] -= 900.0                      # image 1: every row collapses onto its left border pixel(s)
    grads = []
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        ig, fg = img.clone().requires_grad_(True), g(flow).requires_grad_(True)
        (ops.flow_warp(ig, fg) * go).sum().backward()
        grads.append((ig.grad.clone(), fg.grad.clone()))
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0, "the diverged-flow backward must stay bounded"
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    ir, fr = img.cpu().clone().requires_grad_(True), flow.clone().requires_grad_(True)
    (O.resample(ir, fr) * go.cpu()).sum().backward()
    rel_close(grads[0][0], ir.grad, 5e-5, "d image (diverged flow)")
    rel_close(grads[0][1], fr.grad, 2e-4, "d flow (diverged flow)")


@pytest.mark.parametrize("shape", [(40, 16, 4, 8), (5, 64, 32, 64), (2, 3, 128, 256), (3, 7, 9, 13)])
@pytest.mark.parametrize("with_occ", [False, True])
def test_flow_warp_vs_oracle(shape, with_occ):
    N, C, H, W = shape
    img, flow = rnd(1, *shape), rnd(2, N, 2, H, W, scale=2.5)
    occ = (torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(3))) if with_occ else None
    ir, fr = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    yr = O.resample(ir, fr)
    if with_occ:
        yr = yr * occ
    go = rnd(4, *shape)
    (yr * go).sum().backward()
    ig, fg = g(img).requires_grad_(True), g(flow).requires_grad_(True)
    y = ops.flow_warp(ig, fg, g(occ) if with_occ else None)
    (y * g(go)).sum().backward()
    assert torch.equal(y.detach().cpu(), yr.detach()), "forward bit-exact"
    close(ig.grad, ir.grad, 1e-5, 1e-5, "d image")
    close(fg.grad, fr.grad, 1e-4, 1e-4, "d flow")


def test_flow_warp_full_size_c_oracle_bitexact():
    """BASELINE-size image warp against the C restatement (oracle/c2m_oracle_index.c), bit for bit."""
    lib = oracle_build.load()
    N, C, H, W = 5, 3, 128, 256
    img, flow = rnd(1, N, C, H, W).numpy(), rnd(2, N, 2, H, W, scale=3.0).numpy()
    ref = np.zeros_like(img)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oc_resample(p(img), p(flow), None, p(ref), N, C, H, W, oracle_build.FMA_MODE)
    y = ops.flow_warp(g(torch.from_numpy(img)), g(torch.from_numpy(flow))).cpu().numpy()
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("shape", [(2, 3, 4, 6), (5, 8, 16, 32), (1, 2, 7, 5), (1, 1, 1, 4), (2, 2, 3, 2), (3, 2, 9, 14),
                                   # Wi >= 64 and Hi >= 8: the LDS-tiled backward kernel (ragged tiles included)
                                   (2, 3, 9, 70), (1, 2, 16, 64), (1, 1, 8, 130), (3, 2, 21, 67)])
def test_upsample2x(shape):
    x = rnd(1, *shape)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear")
    go = rnd(2, *yr.shape)
    (yr * go).sum().backward()
    xg = g(x).requires_grad_(True)
    y = ops.upsample2x(xg)
    (y * g(go)).sum().backward()
    close(y, yr, 1e-6, 1e-6, "upsample fwd")
    close(xg.grad, xr.grad, 1e-5, 1e-6, "upsample bwd")


@pytest.mark.parametrize("name", names("op_resize_flow"))
def test_resize_flow_golden(name):
    from c2m_amd.utils import resize_flow
    c = Case(name)
    close(resize_flow(g(c.group("in")["flow"]), c.meta["size"]), c.group("out")["y"], 1e-5, 1e-6)


@pytest.mark.parametrize("size", [(8, 16), (4, 8), (16, 32), (5, 9)])
def test_resize_bilinear_align_false(size):
    x = rnd(1, 3, 2, 32, 64)
    close(ops.resize_bilinear(g(x), size), F.interpolate(x, size=size, mode="bilinear"), 1e-5, 1e-6)


@pytest.mark.parametrize("shape", [(3, 5, 8, 12), (2, 3, 7, 9), (1, 2, 6, 5)])   # even sizes: one thread per window
def test_maxpool(shape):
    x = rnd(1, *shape)
    x[0, 0, 0, 0:2] = 1.0   # a tie: gradient must go to the first maximum
    x[0, 0, 1, 0:2] = 1.0
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    go = rnd(2, *yr.shape)
    (yr * go).sum().backward()
    xg = g(x).requires_grad_(True)
    y = ops.maxpool2x2(xg)
    (y * g(go)).sum().backward()
    assert torch.equal(y.detach().cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


# ----------------------------------------------------------------------------------------------- index / mask path
@pytest.mark.parametrize("name", names("op_raster"))
def test_sparse_raster_and_occlusion_golden_bitexact(name):
    c = Case(name)
    i = c.group("in")
    th = i["targets_theta"] if c.meta["use_gt"] else i["targets_theta"] * 1.01
    bw, fw, binm = ops.sparse_raster(g(i["instance"][:, 0]), i["ids"][:, -1], i["batch"], g(th))
    o = c.group("out")
    assert torch.equal(binm.cpu(), c.mask("sparse_motion_bin")), "sparse_motion_bin"
    assert torch.equal(bw.cpu(), o["sparse_motion_bw"]), "sparse_motion_bw"
    assert torch.equal(fw.cpu(), o["sparse_motion_fw"]), "sparse_motion_fw"
    occ_bw = ops.occlusion_splat(fw, want_map=False, want_clip=True)[1]
    occ_fw = ops.occlusion_splat(bw, want_map=False, want_clip=True)[1]
    assert int((occ_bw.cpu() != c.mask("sparse_occ_bw")).sum()) == 0
    assert int((occ_fw.cpu() != c.mask("sparse_occ_fw")).sum()) == 0


@pytest.mark.parametrize("name", names("op_occlusion"))
def test_occlusion_splat_golden_bitexact(name):
    c = Case(name)
    occ, clip = ops.occlusion_splat(g(c.group("in")["flow"]), want_map=True, want_clip=True)
    o = c.group("out")
    assert torch.equal(occ.cpu(), o["y"]), f"{int((occ.cpu() != o['y']).sum())} pixels differ (exact summation order)"
    assert torch.equal(clip.cpu(), o["clip"])


def test_occlusion_splat_full_size_vs_c_oracle_and_converging_flow():
    lib = oracle_build.load()
    B, H, W = 3, 128, 256
    flow = rnd(7, B, 2, H, W, scale=2.0)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flow[2, 0] = (100.0 - xs) * 0.9     # strongly converging field: long per-target lists (slow path)
    flow[2, 1] = (60.0 - ys) * 0.9
    ref = np.zeros((B, 1, H, W), np.float32)
    f = flow.numpy()
    lib.oc_occlusion_splat(f.ctypes.data_as(ctypes.c_void_p), ref.ctypes.data_as(ctypes.c_void_p), B, H, W)
    occ, _ = ops.occlusion_splat(g(flow))
    assert np.array_equal(occ.cpu().numpy(), ref)


# ----------------------------------------------------------------------------------------------- losses
def test_losses_golden():
    c = Case("op_losses")
    i, o, gin = c.group("in"), c.group("out"), c.group("gin")
    a = g(i["a"]).requires_grad_(True)
    b, m = g(i["b"]), g(i["mask"])
    ssim = ops.ssim_loss(O.fold_time(a), O.fold_time(b))
    l1 = ops.l1_mean(a, b)
    l1m = ops.l1_mean(a, b, m)
    (ssim * 1.5 + l1 * 0.7 + l1m * 2.0).backward()
    close(ssim, o["ssim"], 1e-5, 1e-6, "ssim")
    close(l1, o["l1"], 1e-5, 1e-6, "l1")
    close(l1m, o["l1_masked"], 1e-5, 1e-6, "masked l1")
    close(a.grad, gin["a"], 1e-3, 1e-7, "d a")


def test_l1_mean_grad_to_second_argument():
    a, b = rnd(1, 2, 1, 5, 8, 8), rnd(2, 2, 1, 5, 8, 8)
    br = b.clone().requires_grad_(True)
    F.l1_loss(a, br).backward()
    bg = g(b).requires_grad_(True)
    ops.l1_mean(g(a), bg).backward()
    close(bg.grad, br.grad, 1e-6, 1e-9)


def test_roi_align_vs_oracle():
    from oracle import thirdparty
    feat = rnd(1, 4, 6, 32, 64)
    boxes = torch.tensor([[0, 5.3, 30.2, 95.7, 80.1], [1, 0.0, 0.0, 256.0, 128.0], [3, 100.5, 60.0, 101.0, 60.4],
                          [2, 200.0, 100.0, 270.0, 140.0], [0, 20.0, 40.0, 75.0, 85.0]], dtype=torch.float32)
    fr = feat.clone().requires_grad_(True)
    yr = thirdparty.roi_align(fr, boxes, 7, spatial_scale=0.25)
    go = rnd(2, *yr.shape)
    (yr * go).sum().backward()
    fg = g(feat).requires_grad_(True)
    y = ops.roi_align(fg, g(boxes), 7, spatial_scale=0.25)
    (y * g(go)).sum().backward()
    close(y, yr, 1e-4, 1e-5, "roi_align fwd")
    close(fg.grad, fr.grad, 1e-4, 1e-5, "roi_align bwd")


# ----------------------------------------------------------------------------------------------- edge cases
def test_sparse_raster_without_objects_and_with_background_id():
    """No nodes at all, and nodes whose instance id is 0 (the reference skips id 0: dense_motion.py:103-106)."""
    inst = torch.zeros(2, 16, 32)
    inst[1, 4:9, 5:20] = 26001.0
    bw, fw, binm = ops.sparse_raster(g(inst), torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long),
                                     g(torch.zeros(0, 5, 6)))
    assert bw.shape == (2, 2, 5, 16, 32) and not bw.any() and not fw.any() and not binm.any()
    theta = torch.tensor([1.0, 0.0, 0.1, 0.0, 1.0, 0.05]).repeat(2, 5, 1)
    ids, batch = torch.tensor([0, 26001]), torch.tensor([0, 1])
    bw, fw, binm = ops.sparse_raster(g(inst), ids, batch, g(theta))
    assert not binm[0].any() and binm[1].any(), "id 0 must be skipped, the real object rasterised"
    lib = oracle_build.load()
    ref = [np.zeros(s, np.float32) for s in ((2, 2, 5, 16, 32), (2, 2, 5, 16, 32), (2, 1, 5, 16, 32))]
    i64 = lambda t: np.ascontiguousarray(t.numpy().astype(np.int64))
    a_inst, a_ids, a_b, a_th = inst.numpy(), i64(ids), i64(batch), np.ascontiguousarray(theta.numpy())
    scratch = np.zeros(4 * 16 * 32, np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oc_sparse_raster(vp(a_inst), vp(a_ids), vp(a_b), vp(a_th), vp(ref[0]), vp(ref[1]), vp(ref[2]), vp(scratch),
                         2, 2, 5, 16, 32, oracle_build.FMA_MODE)
    assert np.array_equal(bw.cpu().numpy(), ref[0]) and np.array_equal(fw.cpu().numpy(), ref[1])
    assert np.array_equal(binm.cpu().numpy(), ref[2])


def test_occlusion_splat_everything_leaves_the_frame():
    flow = torch.full((2, 2, 8, 16), 1000.0)
    occ, clip = ops.occlusion_splat(g(flow), want_map=True, want_clip=True)
    assert not occ.any() and not clip.any()
    occ0, _ = ops.occlusion_splat(g(torch.zeros(1, 2, 8, 16)))       # zero flow: every pixel lands on itself
    assert torch.equal(occ0.cpu(), torch.ones(1, 1, 8, 16))


def test_roi_align_without_boxes():
    feat = g(rnd(1, 2, 4, 8, 8)).requires_grad_(True)
    y = ops.roi_align(feat, g(torch.zeros(0, 5)), 7, spatial_scale=0.25)
    assert y.shape == (0, 4, 7, 7)
    y.sum().backward()
    assert feat.grad is not None and not feat.grad.any()


@pytest.mark.parametrize("xs,cout,k,pad", [((2, 8, 1, 1), 5, 1, 0), ((3, 4, 1, 1), 6, 3, 1), ((1, 1, 2, 2), 1, 3, 1),
                                            ((1, 3, 1, 40), 2, 3, 1)])
def test_conv_degenerate_extents(xs, cout, k, pad):
    x, w, b = rnd(1, *xs), rnd(2, cout, xs[1], k, k, scale=0.3), rnd(3, cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, padding=pad)
    go = rnd(4, *yr.shape)
    (yr * go).sum().backward()
    xg, wg, bg = (g(t).requires_grad_(True) for t in (x, w, b))
    y = ops.conv(xg, wg, bg, stride=1, padding=pad)
    (y * g(go)).sum().backward()
    rel_close(y, yr, 2e-5, "fwd")
    rel_close(xg.grad, xr.grad, 5e-5, "dgrad")
    rel_close(wg.grad, wr.grad, 1e-4, "wgrad")
    rel_close(bg.grad, br.grad, 1e-4, "bias grad")


@pytest.mark.parametrize("case", [(6, 8, 20, 24, 16, 3, 1, 1, "reflect"), (5, 4, 3, 16, 16, 8, (3, 4, 4), (1, 2, 2), 1, "reflect"),
                                  (7, 64, 16, 32, 64, 3, 1, 1, "zeros")])
def test_conv_batch_chunking_for_tensors_over_2gib(case, monkeypatch):
    """Activations >= 2 GiB (40 folded frames x 128 channels at 256x512) run as batch chunks; the limit is lowered here
    so that small tensors take that path: same outputs and gradients as the single launch."""
    if len(case) == 9:
        n, cin, h, w_, cout, k, stride, pad, mode = case
        xs, ws = (n, cin, h, w_), (cout, cin, k, k)
    else:
        n, cin, t, h, w_, cout, k, stride, pad, mode = case
        xs, ws = (n, cin, t, h, w_), (cout, cin) + tuple(k)
    x, w, b = rnd(1, *xs), rnd(2, *ws) * 0.1, rnd(3, ws[0]) * 0.1
    nd = len(xs) - 2
    s3, p3 = ops._triple(stride, nd), ops._pad3(pad, nd)
    small = 1 << 10                                       # smallest power of two that still fits one image
    while True:
        monkeypatch.setattr(ops, "_MAX_TENSOR_BYTES", small)
        try:
            ops._chunks_for_2gib(xs, ws, s3, p3)
            break
        except ValueError:
            small *= 2
    monkeypatch.undo()
    outs = []
    for limit in (None, small):
        if limit is not None:
            monkeypatch.setattr(ops, "_MAX_TENSOR_BYTES", limit)
            assert ops._chunks_for_2gib(xs, ws, s3, p3) >= 3
        xg, wg, bg = (g(t_).requires_grad_(True) for t_ in (x, w, b))
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        (y * g(rnd(4, *y.shape))).sum().backward()
        outs.append((y.detach(), xg.grad, wg.grad, bg.grad))
    for a, c, name in zip(outs[0], outs[1], ("y", "dx", "dw", "db")):
        rel_close(c, a, 2e-5, name)
    with pytest.raises(ValueError):
        monkeypatch.setattr(ops, "_MAX_TENSOR_BYTES", 1024)
        ops.conv(g(x), g(w), None, stride=stride, padding=pad, padding_mode=mode)


@pytest.mark.parametrize("shape,cout,k,pad,mode", [((2, 34, 5, 16, 32), 32, (3, 3, 3), (1, 1, 1), "reflect"),
                                                   ((3, 20, 12, 24), 16, (3, 3), 1, "zeros")])
def test_conv_dgrad_channels_skips_the_gradient_free_tail(shape, cout, k, pad, mode):
    """final_fuse's input is cat([features (32), rastered sparse motion (2, no grad)]): with dgrad_channels the data
    gradient is computed for the leading channels only (a 32-row GEMM instead of 34 rows on a 64-row tile) and must equal
    the leading channels of the full data gradient; the tail comes back as zeros; the weight gradient is unaffected."""
    keep = shape[1] - 2
    x, w = rnd(31, *shape), rnd(32, cout, shape[1], *k, scale=0.1)
    go = None
    res = []
    for dc in (None, keep):
        xg, wg = g(x).requires_grad_(True), g(w).requires_grad_(True)
        y = ops.conv(xg, wg, None, stride=1, padding=pad, padding_mode=mode, dgrad_channels=dc)
        go = g(rnd(33, *y.shape)) if go is None else go
        (y * go).sum().backward()
        res.append((y.detach(), xg.grad, wg.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    rel_close(res[1][1][:, :keep], res[0][1][:, :keep], 1e-6, "leading channels of the data gradient")
    assert float(res[1][1][:, keep:].abs().max()) == 0.0


@pytest.mark.parametrize("shape,cout,has_next", [((3, 8, 16, 32), 16, True), ((2, 3, 18, 20), 64, True), ((2, 32, 8, 16), 48, False),
                                                 ((1, 5, 7, 9), 6, True)])
def test_conv_relu_tap_matches_the_unfused_chain_bit_for_bit(shape, cout, has_next):
    """ops.conv_relu_tap (perceptual-loss tap of the frozen VGG: relu(conv) feeding the next conv AND an L1 mean) against
    conv(act='relu') + l1_mean + autograd's own gradient sum: same fp32 arithmetic -> identical bits in y, the L1 value and dX.
    has_next = False is the last tap (relu5_1): no gradient arrives through y itself."""
    seed = zlib.crc32(str((shape, cout)).encode()) % 10000
    x0 = rnd(seed, *shape)
    w = g(rnd(seed + 1, cout, shape[1], 3, 3, scale=(1.0 / (shape[1] * 9)) ** 0.5))
    b = g(rnd(seed + 2, cout, scale=0.1))
    w2 = g(rnd(seed + 3, 8, cout, 3, 3, scale=(1.0 / (cout * 9)) ** 0.5))
    res = []
    for fused in (False, True):
        x = g(x0).requires_grad_(True)
        if fused:
            y, l = ops.conv_relu_tap(x, w, b, None if False else _tap_target(seed, w, b, x0))
        else:
            y = ops.conv(x, w, b, stride=1, padding=1, padding_mode="zeros", act="relu")
            l = ops.l1_mean(y, _tap_target(seed, w, b, x0))
        total = l * 3.0
        if has_next:
            total = total + ops.conv(y, w2, None, stride=1, padding=1, padding_mode="zeros").square().mean()
        total.backward()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), l.detach().clone(), x.grad.clone()))
    for a, c, what in zip(res[0], res[1], ("y", "l1", "dX")):
        assert torch.equal(a, c), f"{what}: fused tap differs from the unfused chain"
    assert float(res[1][2].abs().max()) > 0


def _tap_target(seed, w, b, x0):
    with torch.no_grad():
        return ops.conv(g(x0) * 0.9 + 0.05, w, b, stride=1, padding=1, padding_mode="zeros", act="relu")


@pytest.mark.parametrize("shape,cout", [((3, 8, 16, 32), 16), ((2, 3, 18, 20), 64), ((1, 64, 8, 16), 48)])
def test_conv_relu_pool_matches_the_unfused_chain_bit_for_bit(shape, cout):
    """ops.conv_relu_pool (conv -> ReLU -> MaxPool2d of the frozen VGG with the ReLU backward inside the pool backward) against
    conv(act='relu') + maxpool2x2: identical bits in the output and in dX, incl. all-zero windows (dead ReLU regions)."""
    seed = zlib.crc32(str(("crp", shape, cout)).encode()) % 10000
    x0 = rnd(seed, *shape)
    w = g(rnd(seed + 1, cout, shape[1], 3, 3, scale=(1.0 / (shape[1] * 9)) ** 0.5))
    b = g(rnd(seed + 2, cout, scale=0.1) - 0.3)                    # many dead outputs -> windows whose maximum is 0
    res = []
    for fused in (False, True):
        x = g(x0).requires_grad_(True)
        p = ops.conv_relu_pool(x, w, b) if fused else \
            ops.maxpool2x2(ops.conv(x, w, b, stride=1, padding=1, padding_mode="zeros", act="relu"))
        go = g(rnd(seed + 3, *p.shape))
        (p * go).sum().backward()
        torch.cuda.synchronize()
        res.append((p.detach().clone(), x.grad.clone()))
    assert float((res[0][0] == 0).float().mean()) > 0.05, "the case should contain all-zero pooling windows"
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(6, 1, 8, 12), (4, 3, 5, 4), (3, 1, 4, 4), (2, 2, 33, 65)])
def test_reflect_border_add_pad1_kernel_matches_the_general_one_and_autograd(shape, dtype):
    """c2m_reflect_border_add for the 2-D pad-1 case runs a division-free kernel (round 4): same sources in the same order as the
    general kernel -> bit-identical; and both equal the adjoint of F.pad(mode='reflect') restricted to the ring."""
    import os
    import torch.nn.functional as F
    from c2m_amd import _lib
    NC, T, H, W = shape
    L = _lib.lib()
    g = torch.Generator().manual_seed(H * 100 + W)
    tgt = torch.randn(NC, T, H + 2, W + 2, generator=g).to(DEV).to(dtype)
    base = torch.randn(NC, T, H, W, generator=g).to(DEV).to(dtype)
    dt = 1 if dtype == torch.bfloat16 else 0
    outs = []
    for general in (False, True):
        if general:
            os.environ["C2M_FOLD_GENERAL"] = "1"
        try:
            dx = base.clone()
            _lib.check(L.c2m_reflect_border_add(ops._p(tgt), ops._p(dx), NC, T, H, W, 0, 1, 1, dt, ops._stream()), "border add")
            torch.cuda.synchronize()
        finally:
            os.environ.pop("C2M_FOLD_GENERAL", None)
        outs.append(dx)
    assert torch.equal(outs[0], outs[1]), "pad-1 kernel vs general kernel"
    # adjoint of the reflect pad: the ring of tgt folded onto the interior positions it mirrors, interior of tgt NOT added
    x = torch.zeros(NC * T, 1, H, W, dtype=torch.float64, requires_grad=True)
    ring = tgt.double().cpu().reshape(NC * T, 1, H + 2, W + 2).clone()
    ring[:, :, 1:-1, 1:-1] = 0
    F.pad(x, (1, 1, 1, 1), mode="reflect").backward(ring)
    ref = base.double().cpu().reshape(NC * T, 1, H, W) + x.grad
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    close(outs[0].double().cpu().reshape(NC * T, 1, H, W), ref, tol, tol, "border fold vs the adjoint of reflect padding")


@pytest.mark.parametrize("N,H,C,empty_rows", [(24, 4, 512, False), (6, 4, 512, True), (64, 2, 1024, False), (13, 3, 96, True)])
def test_gatv2_dense_attention_one_launch_vs_the_torch_form(N, H, C, empty_rows, monkeypatch):
    """csrc/gnn.hip (the attention of thirdparty.GATv2Conv in one launch forward, two backward) against the layer's own chain of
    torch device ops on the same inputs: multi-edges (multiplicity 2), nodes without incoming edges, forward and the gradients of
    x, lin_l / lin_r, att and the bias; and bit-repeatable (no atomics)."""
    from c2m_amd.thirdparty import GATv2Conv
    g = torch.Generator().manual_seed(100 + N)
    layer = GATv2Conv(C, C, heads=H, concat=False, add_self_loops=False).to(DEV)
    E = 5 * N
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(1 if empty_rows else 0, N, (E,), generator=g)          # node 0 receives nothing when empty_rows
    edge = torch.stack([torch.cat([src, src[:7]]), torch.cat([dst, dst[:7]])]).to(DEV)      # the first 7 edges twice
    x0 = torch.randn(N, C, generator=g).to(DEV)
    w = torch.randn(N, C, generator=g).to(DEV)

    def run(fused):
        monkeypatch.setattr(ops, "_GAT_FUSED", fused)
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = layer(x, edge)
        (y * w).sum().backward()
        return y.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}

    y0, gx0, gp0 = run(False)
    y1, gx1, gp1 = run(True)
    y2, gx2, gp2 = run(True)
    scale = float(y0.abs().max())
    assert (y1 - y0).abs().max() <= 2e-6 * scale + 1e-6, float((y1 - y0).abs().max())
    assert (gx1 - gx0).abs().max() <= 1e-5 * float(gx0.abs().max()) + 1e-7
    for k in gp0:
        assert (gp1[k] - gp0[k]).abs().max() <= 1e-5 * float(gp0[k].abs().max()) + 1e-7, k
    assert torch.equal(y1, y2) and torch.equal(gx1, gx2) and all(torch.equal(gp1[k], gp2[k]) for k in gp1)
    if empty_rows:
        assert torch.equal(y1[0], layer.bias.detach())                  # no incoming edges: the bias alone
