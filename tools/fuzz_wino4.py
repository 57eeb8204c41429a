"""Random 2-D 3x3 stride-1 pad-1 geometries through the FORCED F(4x4,3x3) kernel (forward, data gradient incl. the two-target reflect
form, lrelu epilogue, bit-repeatability) against float64: larger and odder shapes than tools/fuzz_conv.py draws.
    python tools/fuzz_wino4.py [--cases 60] [--seed 0]"""
import argparse, os, sys, random
os.environ["C2M_WINOGRAD"] = "force"
os.environ["C2M_WINO4"] = "force"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from c2m_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
bad = 0
rel = lambda x, r: float((x.detach().cpu().double() - r).abs().max() / max(float(r.abs().max()), 1e-30))
for case in range(a.cases):
    N = rng.choice([1, 2, 3, 5, 6]); Cin = rng.choice([3, 8, 9, 16, 31, 32, 64, 100, 128, 200]); Cout = rng.choice([8, 24, 33, 64, 65, 128, 130, 192])
    H = rng.choice([2, 4, 5, 15, 16, 17, 31, 32, 33, 48, 64, 80]); W = rng.choice([2, 8, 31, 32, 33, 34, 63, 64, 65, 96, 130, 160])
    mode = rng.choice(["zeros", "reflect"])
    g = torch.Generator().manual_seed(9000 + case)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5; b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    xp = F.pad(xr, (1, 1, 1, 1), mode="reflect") if mode == "reflect" else F.pad(xr, (1, 1, 1, 1))
    yr = F.conv2d(xp, wr, br)
    go = torch.randn(*yr.shape, generator=g)
    (yr * go.double()).sum().backward()
    try:
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode)
        pl = ops._plan(xg, wg, (1, 1, 1), (0, 1, 1), mode == "reflect")
        assert pl.wino4_fwd and pl.wino4_dgrad, "not routed to F(4x4,3x3)"
        (y * go.cuda()).sum().backward()
        with torch.no_grad():
            ya = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode, act="lrelu")
            yb = ops.conv(xg, wg, bg, stride=1, padding=1, padding_mode=mode, act="lrelu")
        torch.cuda.synchronize()
        errs = (rel(y, yr.detach()), rel(xg.grad, xr.grad), rel(ya, F.leaky_relu(yr.detach(), 0.2)))
        ok = errs[0] <= 2e-5 and errs[1] <= 5e-5 and errs[2] <= 2e-5 and torch.equal(ya, yb) and bool(torch.isfinite(xg.grad).all())
    except Exception as e:                                   # noqa: BLE001
        ok, errs = False, (repr(e)[:200],)
    if not ok:
        bad += 1
        print("FAIL", (N, Cin, H, W), Cout, mode, errs, flush=True)
print(f"{a.cases} cases, {bad} failures (seed {a.seed})")
sys.exit(1 if bad else 0)
