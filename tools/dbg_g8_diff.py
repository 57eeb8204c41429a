"""Which of the two bf16 paths (NCHW gather / NC8 gather) is off on the shapes where tools/ab_g8.py saw > 1 bf16 ulp between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from c2m_amd import ops
DEV = "cuda:0"
ops.set_conv_precision("bf16")
for xs, Cout, k, stride, pad, mode in [((40, 256, 32, 64), 64, (1, 1), 1, 0, "zeros"), ((40, 512, 8, 16), 512, (3, 3), 1, 1, "zeros"),
                                       ((8, 64, 4, 32, 64), 128, (3, 4, 4), (1, 2, 2), 1, "reflect")]:
    g = torch.Generator().manual_seed(5)
    x = torch.randn(*xs, generator=g).to(DEV).bfloat16()
    taps = 1
    for v in k:
        taps *= v
    w = (torch.randn(Cout, xs[1], *k, generator=g) / (xs[1] * taps) ** 0.5).bfloat16().float().to(DEV)
    nd = len(xs) - 2
    xr = x.float().requires_grad_(True)
    p3 = (pad,) * nd if isinstance(pad, int) else pad
    pads = []
    for p_ in reversed(p3):
        pads += [p_, p_]
    xp = F.pad(xr, pads, mode=mode if mode == "reflect" else "constant") if any(pads) else xr
    yr = (F.conv3d if nd == 3 else F.conv2d)(xp, w, None, stride=stride)
    go = torch.randn(*yr.shape, generator=g).to(DEV).bfloat16()
    yr.backward(go.float())
    for g8 in (False, True):
        ops._G8 = g8
        ops._geom_cache.clear()
        xb = x.clone().requires_grad_(True)
        y = ops.conv(xb, w, None, stride=stride, padding=pad, padding_mode=mode)
        y.backward(go)
        ey = float((y.float() - yr).abs().max() / yr.abs().max())
        eg = float((xb.grad.float() - xr.grad).abs().max() / xr.grad.abs().max())
        bad = ((xb.grad.float() - xr.grad).abs() > 0.01 * xr.grad.abs().max()).nonzero()
        print(xs, Cout, k, "g8" if g8 else "old", f"fwd err {ey:.1e} dgrad err {eg:.1e} bad {bad.shape[0]}", bad[:4].tolist(), flush=True)
