"""Round-3 hunt: conv_thin_wgrad_rows_kernel on a side stream next to the bf16 data-gradient kernels of the same layer (7x7
RGB head, bf16 mode): slab entries differ from the sequential run when conv_igemm.hip is built WITH the SLP vectoriser; which of the
data-gradient launches disturbs it (answer: only c2m_conv_igemm's bf16 gather kernel).  c2m_amd/build.py builds that file without SLP."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from c2m_amd import ops, _lib
ops.set_conv_precision("bf16")
dev = torch.device("cuda:0")
L = _lib.lib()
xs, cout, k, pad = (40, 32, 128, 256), 3, (7, 7), 3
g = torch.Generator().manual_seed(1)
x = torch.randn(*xs, generator=g).to(dev)
w = (torch.randn(cout, xs[1], *k, generator=g) / 40).to(dev)
gy = torch.randn(40, 3, 128, 256, generator=g).to(dev).to(torch.bfloat16)
pl = ops._plan(x, w, (1, 1, 1), (0, 3, 3), True)
side = ops._side_stream(dev)
main = torch.cuda.current_stream()
def wgrad_on(stream, gyw, slab, gw, gb):
    with torch.cuda.stream(stream):
        pl.wg_geom[ops.G.X_BYTES], pl.wg_geom[ops.G.DY_BYTES] = x.numel() * 4, gyw.numel() * 4
        pl.wg_geom[ops.G.X_TYPE] = 0
        _lib.check(L.c2m_conv_wgrad(ops._p(gyw), ops._p(x), ops._p(slab), ops._p(gw), ops._p(gb), ops._p(pl.wg_tab),
                                    ops._gp(pl.wg_geom), ops._stream()), "wgrad")
gyw = gy.float().contiguous()
nsl = pl.wg_splits * cout * pl.J
def fresh():
    return torch.full((nsl,), 7.0, device=dev), torch.empty_like(w), torch.empty(cout, device=dev)
slab0, gw0, gb0 = fresh()
wgrad_on(main, gyw, slab0, gw0, gb0)
torch.cuda.synchronize()
print("J", pl.J, "splits", pl.wg_splits, "slab floats", nsl)
for it in range(3):
    slab1, gw1, gb1 = fresh()
    gyw1 = gyw.clone(); x_before = x.clone()
    torch.cuda.synchronize()
    side.wait_stream(main)
    wgrad_on(side, gyw1, slab1, gw1, gb1)
    gx = ops._conv_dgrad(pl, w, gy, False, torch.float32)
    torch.cuda.synchronize()
    d = (slab1 != slab0)
    print(it, "gw eq", torch.equal(gw1, gw0), "gyw intact", torch.equal(gyw1, gyw), "x intact", torch.equal(x, x_before),
          "slab differing", int(d.sum()))
    if d.any():
        idx = d.nonzero().flatten()
        per = cout * pl.J
        sp = idx // per; m = (idx % per) // pl.J; col = idx % pl.J
        print("   splits", sorted(set(sp.tolist()))[:20], "m", sorted(set(m.tolist())), "cols", sorted(set(col.tolist()))[:24])
        print("   got", slab1[idx[:6]].tolist(), "want", slab0[idx[:6]].tolist())

print("---- which dgrad kernel disturbs the concurrent thin wgrad")
class Skip:
    def __init__(self, names): self.names = names; self.real = {}
    def __enter__(self):
        for n in self.names:
            self.real[n] = getattr(L, n)
            setattr(L, n, (lambda *a, **k: 0))
    def __exit__(self, *a):
        for n, f in self.real.items(): setattr(L, n, f)
folds = ["c2m_reflect_border_add", "c2m_reflect_fold"]
for label, names in (("all kernels", []), ("no folds", folds), ("no igemm", ["c2m_conv_igemm"]),
                     ("no igemm, no folds (pack + allocs only)", ["c2m_conv_igemm"] + folds), ("no splitk", ["c2m_splitk_reduce"])):
    res = []
    with Skip(names):
        for it in range(3):
            slab1, gw1, gb1 = fresh()
            torch.cuda.synchronize()
            side.wait_stream(main)
            wgrad_on(side, gyw, slab1, gw1, gb1)
            gx = ops._conv_dgrad(pl, w, gy, False, torch.float32)
            torch.cuda.synchronize()
            res.append(int((slab1 != slab0).sum()))
    print(label, "-> differing slab entries", res)
print("dgrad splits", pl.dgrad_splits, "classes", [(c["taps"], c["npix"], c["patch"]) for c in pl.classes], "needs_zero", pl.dgrad_needs_zero)
