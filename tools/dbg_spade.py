import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from c2m_amd import ops
dev = "cuda:0"
def rel(x, y):
    x, y = x.detach().cpu().double(), y.detach().double()
    return float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
def run(shape, act, kind, sc=1.0, off=0.0):
    g = torch.Generator().manual_seed(7)
    C = shape[1]
    x = torch.randn(*shape, generator=g) * sc + off; go = torch.randn(*shape, generator=g)
    xr, xg = x.double().requires_grad_(True), x.to(dev).requires_grad_(True)
    if kind == "spade":
        gb = torch.randn(shape[0], 2 * C, *shape[2:], generator=g) * 0.5
        gbr = gb.double().requires_grad_(True); ga, be = gbr.chunk(2, 1)
        yr = F.instance_norm(xr) * (1 + ga) + be
        gbg = gb.to(dev).requires_grad_(True)
        y = ops.spade_norm_act(xg, gbg, act=act)
    else:
        yr = F.instance_norm(xr); y = ops.instance_norm_act(xg, act=act)
    if act == "lrelu": yr = F.leaky_relu(yr, 0.2)
    (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
    out = [rel(y, yr), rel(xg.grad, xr.grad)]
    if kind == "spade": out.append(rel(gbg.grad, gbr.grad))
    return ["%.1e" % e for e in out]
for shape in [(1, 1, 1, 16384), (1, 1, 1, 8448), (1, 1, 1, 8192), (1, 1, 1, 8196), (3, 33, 1, 8448), (1, 2, 128, 256), (1, 1, 1, 32768), (2, 3, 1, 9000)]:
    for sc, off in [(1.0, -50.0), (1.0, 50.0), (1.0, 5.0)]:
        print(shape, "in", "scale", sc, "offset", off, run(shape, "lrelu", "in", sc, off), run(shape, None, "in", sc, off))
