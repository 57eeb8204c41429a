"""One conv geometry against float64 (forward, dx, dw, db), default routing.  python tools/dbg_conv_case.py "N,C,T,H,W" Cout "kt,kh,kw" "st,sh,sw" "pt,ph,pw" mode"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from c2m_amd import ops
xs = tuple(int(v) for v in sys.argv[1].split(",")); cout = int(sys.argv[2])
ks = tuple(int(v) for v in sys.argv[3].split(",")); st = tuple(int(v) for v in sys.argv[4].split(","))
pad = tuple(int(v) for v in sys.argv[5].split(",")); mode = sys.argv[6]
g = torch.Generator().manual_seed(7)
x = torch.randn(*xs, generator=g); w = torch.randn(cout, xs[1], *ks, generator=g) / (xs[1] * float(torch.tensor(ks).prod())) ** 0.5
b = torch.randn(cout, generator=g) * 0.1
xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
xp = xr
if mode == "reflect":
    tup = []
    for p in reversed(pad):
        tup += [p, p]
    xp = F.pad(xr, tuple(tup), mode="reflect"); pp = (0,) * len(pad)
else:
    pp = pad
yr = (F.conv3d if len(ks) == 3 else F.conv2d)(xp, wr, br, stride=st, padding=pp)
go = torch.randn(*yr.shape, generator=g)
(yr * go.double()).sum().backward()
xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
y = ops.conv(xg, wg, bg, stride=st, padding=pad, padding_mode=mode)
(y * go.cuda()).sum().backward()
rel = lambda a, r: float((a.cpu().double() - r).abs().max() / r.abs().max())
print(sys.argv[1:], "y %.2e dx %.2e dw %.2e" % (rel(y, yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad)))
d = (xg.grad.cpu().double() - xr.grad).abs()
idx = (d > 1e-3 * xr.grad.abs().max()).nonzero()
print("bad elements:", idx.shape[0], "of", d.numel(), "; t values:", sorted(set(idx[:, 2].tolist())) if idx.numel() else [], "h:", sorted(set(idx[:, 3].tolist()))[:12] if idx.numel() else [], "w:", sorted(set(idx[:, 4].tolist()))[:12] if idx.numel() and len(xs) == 5 else [])
