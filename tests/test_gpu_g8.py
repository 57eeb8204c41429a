"""The NC8 gather form of the bf16 forward / data-gradient kernels (conv_gather_nc8_kernel, conv_igemm.hip; round 4) against an fp32
convolution of the same bf16-representable operands (every product exact in fp32: only the summation order differs): the layers the
NC8 patch forms do not take -- (3,4,4) / (4,4,4) stride-2 3-D blocks, stride-2 2-D layers on small or odd maps, 3x3 on small maps,
1x1, 7x7 --, stride-parity classes (batched and per class), the two-target reflect epilogue, split-K, ragged row / pixel tiles,
channel counts off the 8 / 16 grids, every tile variant, NaNs planted behind the tensors, the one-launch pack refresh.  Every case
asserts that the plan really routes to the gather form."""
import pytest
import torch
import torch.nn.functional as F

from c2m_amd import ops
from gpu_util import rel_close, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(t):
    return t.bfloat16().float()


@pytest.fixture(autouse=True)
def _bf16_mode():
    prev = ops.set_conv_precision("bf16")
    v0 = ops._G8_VARIANT
    ops._geom_cache.clear()
    yield
    ops._G8_VARIANT = v0
    ops._geom_cache.clear()
    ops.set_conv_precision(prev)


def _ref(x, w, b, stride, pad, mode):
    nd = x.dim() - 2
    pads = []
    for p_ in reversed(pad[3 - nd:]):
        pads += [p_, p_]
    xp = F.pad(x, pads, mode="reflect") if (mode == "reflect" and any(pads)) else F.pad(x, pads)
    conv = F.conv3d if nd == 3 else F.conv2d
    return conv(xp, w, b, stride=stride[3 - nd:])


# xshape, Cout, kernel, stride, pad, mode
CASES = [
    ((2, 32, 16, 16), 64, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect"),      # stride 2 on a small map: four batched parity classes
    ((1, 40, 16, 20), 100, (1, 4, 4), (1, 2, 2), (0, 1, 1), "zeros"),       # ragged rows (100 = 64 + 36), 40 channels = 2.5 chunks
    ((2, 48, 8, 8), 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), "reflect"),        # 3x3 on a map the 8 x 32 patch tiles would waste
    ((1, 16, 3, 16, 16), 40, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),   # 3-D down block, (1,2,2) stride
    ((1, 24, 4, 8, 16), 64, (4, 4, 4), (2, 2, 2), (1, 1, 1), "reflect"),    # (4,4,4) stride 2: eight classes
    ((1, 24, 4, 8, 16), 36, (4, 4, 4), (2, 2, 2), (1, 1, 1), "zeros"),
    ((2, 64, 8, 16), 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), "zeros"),         # 1x1
    ((1, 16, 16, 24), 12, (1, 7, 7), (1, 1, 1), (0, 3, 3), "reflect"),      # 7x7, 49 taps
    ((1, 256, 8, 8), 136, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect"),      # deep reduction, few pixels: split-K
    ((3, 24, 3, 8, 8), 200, (3, 3, 3), (1, 1, 1), (1, 1, 1), "zeros"),      # 3x3x3 below the patch form's map size; 128-row tile + ragged second tile
    ((2, 16, 16, 16), 16, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect"),      # 4 K-steps per class: no split, two-target epilogue
    ((1, 32, 3, 32, 64), 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),   # a 3-D down block of the motion encoder at model size
    ((2, 32, 17, 17), 48, (1, 4, 4), (1, 2, 2), (0, 1, 1), "zeros"),        # odd map: four classes of different extents, one launch each
    ((2, 32, 17, 16), 48, (1, 4, 4), (1, 2, 2), (0, 1, 1), "zeros"),        # two runs of two classes
]


def _run_case(case, dgrad_rows=None):
    xs, Cout, k, stride, pad, mode = case
    nd = len(xs) - 2
    Cin = xs[1]
    taps = k[0] * k[1] * k[2]
    x = _bf(rnd(61, *xs))
    w = _bf(rnd(62, Cout, Cin, *k[3 - nd:], scale=(1.0 / (Cin * taps)) ** 0.5))
    b = rnd(63, Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = _ref(xr, wr, br, stride, pad, mode)
    go = _bf(rnd(64, *yr.shape))
    (yr * go).sum().backward()
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    pl = ops._plan(xg.bfloat16(), wg, stride, pad, mode == "reflect", dgrad_rows)
    y = ops.conv(xg, wg, bg, stride=stride[3 - nd:] if nd == 3 else stride[1:], padding=pad[3 - nd:] if nd == 3 else pad[1:],
                 padding_mode=mode)
    (y.float() * go.to(DEV)).sum().backward()
    return pl, y, yr, xg, xr, wg, wr, bg, br


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[0]) + f"-{c[1]}-k{''.join(map(str, c[2]))}-{c[5]}")
def test_gather_nc8_forward_and_data_gradient(case):
    pl, y, yr, xg, xr, wg, wr, bg, br = _run_case(case)
    plane = 1
    for v in case[0][2:]:
        plane *= v
    # (the layout pass takes planes of whole 8-element groups: the forward of an odd map stays on the NCHW gather kernel)
    assert pl.g8_dgrad and pl.g8_fwd == (plane % 8 == 0), "forward and data gradient must run on the NC8 gather form"
    assert y.dtype == torch.bfloat16
    rel_close(y.float(), yr, 4e-3, "NC8 gather forward (one RNE rounding of the exact sum)")
    rel_close(xg.grad, xr.grad, 5e-5, "NC8 gather data gradient")
    rel_close(wg.grad, wr.grad, 1e-4, "weight gradient")
    rel_close(bg.grad, br.grad, 1e-4, "bias gradient")


def test_gather_nc8_splits_and_classes_are_exercised():
    """The case list above must cover split-K launches, batched classes and per-class launches."""
    seen = dict(split_fwd=False, split_dgrad=False, batched=False, per_class=False, two_target=False)
    for xs, Cout, k, stride, pad, mode in CASES:
        x = torch.empty(xs, device=DEV, dtype=torch.bfloat16)
        w = torch.empty((Cout, xs[1]) + tuple(k[3 - (len(xs) - 2):]), device=DEV)
        pl = ops._plan(x, w, stride, pad, mode == "reflect")
        seen["split_fwd"] |= pl.g8_fwd and pl.g8_fwd_splits > 1
        seen["split_dgrad"] |= pl.g8_dgrad_splits > 1
        seen["batched"] |= pl.cls_batch is not None
        seen["per_class"] |= pl.cls_batch is None and len(pl.classes) > 1
        seen["two_target"] |= bool(pl.reflect and any(pl.pad) and pl.g8_dgrad_splits == 1)
    assert all(seen.values()), seen


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7])
def test_gather_nc8_tile_variants(variant):
    """Every (rows, K-steps per stage, stages) instantiation on a stride-2 layer with ragged rows and a ragged pixel tile."""
    ops._G8_VARIANT = variant
    ops._geom_cache.clear()
    Cout = 24 if variant in (1, 7) else 168
    case = ((3, 40, 12, 16), Cout, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect")
    pl, y, yr, xg, xr, wg, wr, bg, br = _run_case(case)
    assert pl.g8_fwd and pl.g8_dgrad and int(pl.g8_fwd_geom[ops.G.G8_VARIANT]) == variant
    rel_close(y.float(), yr, 4e-3, "forward")
    rel_close(xg.grad, xr.grad, 5e-5, "data gradient")


def test_gather_nc8_bf16_results_and_gradient_free_tail():
    """bf16 target of the data gradient (the type it has inside the network) and dgrad_channels < Cin (a concatenated input whose
    tail carries no gradient: those channels come back as zeros)."""
    xs, Cout, k, stride, pad, mode = ((2, 32, 16, 16), 64, (1, 4, 4), (1, 2, 2), (0, 1, 1), "reflect")
    x, w = _bf(rnd(71, *xs)), _bf(rnd(72, Cout, xs[1], 4, 4, scale=(1.0 / (xs[1] * 16)) ** 0.5))
    xr, wr = x.clone().requires_grad_(True), w.clone()
    yr = _ref(xr, wr, None, stride, pad, mode)
    go = _bf(rnd(73, *yr.shape))
    (yr * go).sum().backward()
    xb = x.to(DEV).bfloat16().requires_grad_(True)
    yb = ops.conv(xb, w.to(DEV), None, stride=2, padding=1, padding_mode=mode)
    yb.backward(go.to(DEV).bfloat16())
    assert xb.grad.dtype == torch.bfloat16
    rel_close(xb.grad.float(), xr.grad, 8e-3, "data gradient, bf16 result (rounded before and after the reflect fold)")
    # a stride-1 layer whose input is a concatenation with a gradient-free tail (dgrad_channels: stride-1 layers only)
    x3, w3 = _bf(rnd(74, 2, 48, 8, 8)).to(DEV).bfloat16(), _bf(rnd(75, 72, 48, 3, 3, scale=0.05)).to(DEV)
    go3 = _bf(rnd(76, 2, 72, 8, 8)).to(DEV).bfloat16()
    xa, xb2 = x3.clone().requires_grad_(True), x3.clone().requires_grad_(True)
    pl = ops._plan(xb2, w3, (1, 1, 1), (0, 1, 1), True, 20)
    assert pl.g8_dgrad and pl.dM == 20
    ops.conv(xa, w3, None, stride=1, padding=1, padding_mode="reflect").backward(go3)
    ops.conv(xb2, w3, None, stride=1, padding=1, padding_mode="reflect", dgrad_channels=20).backward(go3)
    assert torch.equal(xb2.grad[:, :20], xa.grad[:, :20]) and not xb2.grad[:, 20:].any()


def test_gather_nc8_never_reads_past_its_input():
    """The DMAs carry the channel block in a scalar offset the hardware does not range-check: 40 channels = 5 blocks, the sixth
    (second half of the last 16-channel chunk) must read zeros, not the NaNs planted behind the tensor."""
    N, Cin, H, W, Cout = 2, 40, 8, 8, 48
    big = torch.full((N * Cin * H * W + 4096,), float("nan"), device=DEV, dtype=torch.bfloat16)
    x = _bf(rnd(81, N, Cin, H, W))
    big[:x.numel()] = x.to(DEV).bfloat16().reshape(-1)
    xv = big[:x.numel()].view(N, Cin, H, W)
    w = _bf(rnd(82, Cout, Cin, 4, 4, scale=0.05))
    pl = ops._plan(xv, w.to(DEV), (1, 2, 2), (0, 1, 1), True)
    assert pl.g8_fwd
    # the NC8 copy is its own allocation: plant NaNs behind it by building it inside a larger buffer
    xn_big = torch.full((N * 5 * H * W * 8 + 8192,), float("nan"), device=DEV, dtype=torch.bfloat16)
    xn = xn_big[:N * 5 * H * W * 8].view(N, 5, H, W, 8)
    xn.copy_(ops._to_nc8(xv))
    xv._c2m_nc8 = (xv._version, xn)
    y = ops.conv(xv, w.to(DEV), None, stride=2, padding=1, padding_mode="reflect")
    yr = _ref(x, w, None, (1, 2, 2), (0, 1, 1), "reflect")
    assert torch.isfinite(y).all()
    rel_close(y.float(), yr, 4e-3, "forward next to NaNs")


def test_gather_pack_is_refreshed_in_place_by_the_optimizer_step():
    """The bf16 gather images are registered packs (job type 2): after an in-place weight update one c2m_pack_multi launch rebuilds
    them bit-identically to a fresh pack."""
    Cout, Cin = 40, 32
    w = torch.nn.Parameter(_bf(rnd(91, Cout, Cin, 4, 4, scale=0.05)).to(DEV))
    x = _bf(rnd(92, 2, Cin, 8, 8)).to(DEV).bfloat16()
    y0 = ops.conv(x, w, None, stride=2, padding=1, padding_mode="reflect")
    with torch.no_grad():
        w.add_(_bf(rnd(93, Cout, Cin, 4, 4, scale=0.01)).to(DEV))
    n = ops.refresh_trainable_packs([w])
    assert n >= 1
    y1 = ops.conv(x, w, None, stride=2, padding=1, padding_mode="reflect")
    fresh = ops._pack_bf16_gather(w.detach(), Cout, Cin, (1, 4, 4), (1, 1, 1), Cin * 16, 16)
    hit = ops._frozen_pack_cache[(id(w), ("fwd-bf16-g8",))]
    assert torch.equal(hit[3], fresh) and not torch.equal(y0, y1)
