"""Random-shape parity sweep of the element-parallel / reduction ops (norm + activation, x2 up-sampling, 2x2 max-pool, bilinear
resize, masked L1) against torch's own float64 CPU implementations: odd extents, sizes around the kernels' vector / chunk / tile
boundaries (8192-element statistics chunks, 4-wide vector paths, 64- and 32-column up-sampling tiles), 4-D and 5-D tensors.
    python tools/fuzz_ops.py [--cases 200] [--seed 0]
"""
import argparse, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from c2m_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
dev = "cuda:0"
bad = 0


def rel(x, y, floor=1e-30):
    x, y = x.detach().cpu().double(), y.detach().double()
    return float((x - y).abs().max()) / max(float(y.abs().max()), floor)


def act_ref(y, act):
    return F.leaky_relu(y, 0.2) if act == "lrelu" else (F.relu(y) if act == "relu" else y)


def well_conditioned(go, pre, act):
    """The activation's slope switches at pre-activation 0: an fp32 mean that differs from the float64 one in its last bits
    flips the mask of elements within ~|offset| * 1e-6 of 0 (a whole gradient term per flipped element, in ANY fp32
    implementation).  Those elements get no upstream gradient on either side, so the comparison stays a kernel check."""
    return go if act is None else go * (pre.detach().abs() > 1e-3).to(go.dtype)


def check(name, shape, pairs, tol):
    global bad
    # a pair may carry a floor for its scale: a per-channel sum of n random terms is measured against sqrt(n), not against
    # its own (possibly cancelled) value
    errs = [rel(*p) for p in pairs]
    if any(e > tol for e in errs) or not all(bool(torch.isfinite(p[0]).all()) for p in pairs):
        bad += 1
        print("FAIL", name, shape, ["%.2e" % e for e in errs])


def rshape():
    N = rng.choice([1, 2, 3, 5])
    C = rng.choice([1, 3, 4, 7, 16, 32, 33])
    if rng.random() < 0.2:
        return (N, C, rng.choice([1, 2, 5]), rng.choice([4, 9, 16, 33]), rng.choice([8, 17, 64, 130]))
    return (N, C, rng.choice([2, 5, 8, 16, 33, 64, 90, 128]), rng.choice([4, 8, 17, 32, 64, 100, 130, 256]))


for case in range(a.cases):
    g = torch.Generator().manual_seed(5000 + case)
    kind = rng.choice(["bn", "in", "spade", "up", "pool", "resize", "l1"])
    try:
        if kind in ("bn", "in", "spade"):
            shape = rshape()
            C = shape[1]
            act = rng.choice([None, "lrelu", "relu"])
            x = torch.randn(*shape, generator=g) * rng.choice([0.1, 1.0, 30.0]) + rng.choice([0.0, 3.0, -50.0])
            go = torch.randn(*shape, generator=g)
            xr = x.double().requires_grad_(True)
            xg = x.to(dev).requires_grad_(True)
            if kind == "bn":
                nper = shape[0] * int(torch.tensor(shape[2:]).prod())
                if nper < 2:
                    continue
                gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
                gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
                rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
                pre = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
                yr, go = act_ref(pre, act), well_conditioned(go, pre.float(), act)
                gg, bg = gam.to(dev).requires_grad_(True), bet.to(dev).requires_grad_(True)
                rmg, rvg = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                y = ops.batch_norm_act(xg, gg, bg, rmg, rvg, act=act)
                (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
                check("bn/" + str(act), shape, [(y, yr), (xg.grad, xr.grad), (gg.grad, gr.grad, nper ** 0.5), (bg.grad, br.grad, nper ** 0.5), (rmg, rm), (rvg, rv)], 2e-4)
            elif kind == "in":
                if int(torch.tensor(shape[2:]).prod()) < 2:
                    continue
                affine = rng.random() < 0.5
                gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
                gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
                pre = F.instance_norm(xr, None, None, gr if affine else None, br if affine else None, True, 0.1, 1e-5)
                yr, go = act_ref(pre, act), well_conditioned(go, pre.float(), act)
                gg, bg = gam.to(dev).requires_grad_(True), bet.to(dev).requires_grad_(True)
                y = ops.instance_norm_act(xg, gg if affine else None, bg if affine else None, act=act)
                (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
                pairs = [(y, yr), (xg.grad, xr.grad)] + ([(gg.grad, gr.grad, x[:, 0].numel() ** 0.5), (bg.grad, br.grad, x[:, 0].numel() ** 0.5)] if affine else [])
                check("in/" + str(act), shape, pairs, 2e-4)
            else:
                if len(shape) != 4 or shape[2] * shape[3] < 2:
                    continue
                gb = torch.randn(shape[0], 2 * C, *shape[2:], generator=g) * 0.5
                gbr = gb.double().requires_grad_(True)
                gam_, bet_ = gbr.chunk(2, 1)
                pre = F.instance_norm(xr, None, None, None, None, True, 0.1, 1e-5) * (1 + gam_) + bet_
                yr, go = act_ref(pre, act), well_conditioned(go, pre.float(), act)
                gbg = gb.to(dev).requires_grad_(True)
                y = ops.spade_norm_act(xg, gbg, act=act)
                (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
                check("spade/" + str(act), shape, [(y, yr), (xg.grad, xr.grad), (gbg.grad, gbr.grad)], 2e-4)
        elif kind == "up":
            shape = (rng.choice([1, 2, 3]), rng.choice([1, 3, 8, 17]), rng.choice([1, 2, 7, 8, 9, 16, 33, 64]), rng.choice([1, 3, 4, 31, 32, 33, 63, 64, 65, 96, 128]))
            x = torch.randn(*shape, generator=g)
            xr, xg = x.double().requires_grad_(True), x.to(dev).requires_grad_(True)
            yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
            y = ops.upsample2x(xg)
            go = torch.randn(*yr.shape, generator=g)
            (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
            check("upsample2x", shape, [(y, yr), (xg.grad, xr.grad)], 2e-6)
        elif kind == "pool":
            shape = (rng.choice([1, 2, 3]), rng.choice([1, 3, 8, 17]), rng.choice([2, 3, 7, 8, 16, 33, 64]), rng.choice([2, 3, 4, 9, 32, 33, 64, 130]))
            x = torch.randn(*shape, generator=g)
            xr, xg = x.double().requires_grad_(True), x.to(dev).requires_grad_(True)
            yr = F.max_pool2d(xr, 2, 2)
            y = ops.maxpool2x2(xg)
            go = torch.randn(*yr.shape, generator=g)
            (yr * go.double()).sum().backward(); (y * go.to(dev)).sum().backward()
            check("maxpool2x2", shape, [(y, yr), (xg.grad, xr.grad)], 1e-7)
        elif kind == "resize":
            shape = (rng.choice([1, 2]), rng.choice([1, 2, 5]), rng.choice([4, 8, 16, 33]), rng.choice([4, 8, 32, 65]))
            size = (rng.choice([2, 4, 7, 16, 64]), rng.choice([3, 8, 16, 33, 128]))
            ac = rng.random() < 0.5
            x = torch.randn(*shape, generator=g)
            yr = F.interpolate(x.double(), size=size, mode="bilinear", align_corners=ac)
            y = ops.resize_bilinear(x.to(dev), size, align_corners=ac)
            check("resize_bilinear", shape + size, [(y, yr)], 1e-5)      # fp32 source coordinates: a few ulp of the lerp weight
        else:
            shape = rshape()
            if len(shape) != 4:
                shape = shape[:2] + (shape[2] * shape[3], shape[4])
            x, t = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
            use_mask = rng.random() < 0.5
            m = (torch.rand(shape[0], 1, *shape[2:], generator=g) > 0.4).float() if use_mask else None
            xr = x.double().requires_grad_(True)
            lr = F.l1_loss(xr * m.double(), t.double() * m.double()) if use_mask else F.l1_loss(xr, t.double())
            xg = x.to(dev).requires_grad_(True)
            l = ops.l1_mean(xg, t.to(dev), None if m is None else m.to(dev))
            (lr * 3.0).backward(); (l * 3.0).backward()
            check("l1_mean", shape, [(l, lr), (xg.grad, xr.grad)], 2e-6)
        torch.cuda.synchronize()
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print("EXC ", kind, repr(e)[:300])
print(f"{a.cases} cases, {bad} failures (seed {a.seed})")
sys.exit(1 if bad else 0)
