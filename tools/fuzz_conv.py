"""Random-geometry parity sweep of ops.conv (forward, data gradient, weight / bias gradient) against torch's CPU convolution in
float64: shapes the hand-written lists of tests/test_gpu_ops.py do not enumerate (odd extents, channel counts around the tile
sizes, every kernel family incl. the Winograd region shapes, the 3x3x3 temporal pairs, the stride-2 wide bf16 weight gradient).
    python tools/fuzz_conv.py [--cases 150] [--seed 0] [--bf16]
Exit code 1 if any case misses its tolerance (fp32 mode: 2e-5 / 5e-5 / 1e-4 of the tensor scale, like the unit tests).
"""
import argparse, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from c2m_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
rng = random.Random(a.seed)
dev = "cuda:0"
ops.set_conv_precision("bf16" if a.bf16 else "fp32")


def bf(t):
    return t.bfloat16().float() if a.bf16 else t


def ref_conv(x, w, b, stride, pad, mode):
    nd = x.dim() - 2
    if mode == "reflect" and any(pad):
        tup = []
        for p_ in reversed(pad):
            tup += [p_, p_]
        x = F.pad(x, tuple(tup), mode="reflect")
        pad = (0,) * nd
    return (F.conv2d if nd == 2 else F.conv3d)(x, w, b, stride=stride, padding=pad)


def rel(a_, b_):
    a_, b_ = a_.detach(), b_.detach()
    return float((a_ - b_).abs().max()) / max(float(b_.abs().max()), 1e-30)


bad = 0
for case in range(a.cases):
    nd = 3 if rng.random() < 0.25 else 2
    fam = rng.choice(["3x3", "3x3", "3x3", "4x4s2", "4x4s2", "7x7", "1x1", "3x3s2"])
    cin = rng.choice([3, 5, 8, 16, 24, 32, 33, 40, 64, 72, 96, 130])
    cout = rng.choice([1, 3, 4, 8, 16, 31, 32, 48, 64, 80, 128, 136])
    if fam == "3x3":
        k, s, p_ = 3, 1, 1
    elif fam == "3x3s2":
        k, s, p_ = 3, 2, 1
    elif fam == "4x4s2":
        k, s, p_ = 4, 2, 1
    elif fam == "7x7":
        k, s, p_ = 7, 1, 3
        cin, cout = min(cin, 32), min(cout, 32)
    else:
        k, s, p_ = 1, 1, 0
    H = rng.choice([6, 8, 9, 10, 12, 16, 18, 24, 32, 34, 40, 64])
    W = rng.choice([8, 11, 12, 16, 20, 32, 34, 48, 64, 66, 96, 128])
    if fam == "7x7":
        H, W = max(H, 8), max(W, 8)
    N = rng.choice([1, 2, 3])
    mode = rng.choice(["zeros", "reflect"]) if p_ else "zeros"
    if nd == 3:
        T = rng.choice([1, 2, 3, 5])
        kt = rng.choice([1, 3]) if k != 4 else rng.choice([3, 4])
        if kt * k * k > 128:                                  # 3x7x7: more than the 128 taps the kernels' tables hold (raises)
            kt = 1
        st = 2 if kt == 4 else 1
        pt = 1 if kt >= 3 else 0
        if mode == "reflect" and pt >= T:
            mode = "zeros"
        if (T + 2 * pt - kt) // st + 1 <= 0:
            continue
        xs, ks, stride, pad = (N, cin, T, H, W), (kt, k, k), (st, s, s), (pt, p_, p_)
        cin, cout = min(cin, 48), min(cout, 64)
        xs = (N, cin, T, H, W)
    else:
        xs, ks, stride, pad = (N, cin, H, W), (k, k), (s, s), (p_, p_)
    if any((d + 2 * q - kk) // ss + 1 <= 0 for d, q, kk, ss in zip(xs[2:], pad, ks, stride)):
        continue
    if mode == "reflect" and any(q >= d for d, q in zip(xs[2:], pad)):
        mode = "zeros"
    g = torch.Generator().manual_seed(1000 + case)
    x = bf(torch.randn(*xs, generator=g))
    w = bf(torch.randn(cout, cin, *ks, generator=g) / (cin * float(torch.tensor(ks).prod())) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = ref_conv(xr, wr, br, stride, pad, mode)
    go = bf(torch.randn(*yr.shape, generator=g))
    (yr * go.double()).sum().backward()
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    try:
        y = ops.conv(xg, wg, bg, stride=stride, padding=pad, padding_mode=mode)
        (y.float() * go.to(dev)).sum().backward()
        torch.cuda.synchronize()
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print("EXC ", xs, cout, ks, stride, pad, mode, repr(e)[:200])
        continue
    out_bf16 = a.bf16 and y.dtype == torch.bfloat16
    tol = (4e-3 if out_bf16 else 2e-5, 5e-5, 1e-4, 1e-4)
    errs = (rel(y.float().cpu().double(), yr.detach()), rel(xg.grad.cpu().double(), xr.grad), rel(wg.grad.cpu().double(), wr.grad),
            float((bg.grad.cpu().double() - br.grad).abs().max()) / max(float(go.abs().sum()) / cout * 1e-3, float(br.grad.abs().max()), 1e-30))
    if any(e > t for e, t in zip(errs, tol)) or not all(torch.isfinite(t_).all() for t_ in (y, xg.grad, wg.grad, bg.grad)):
        bad += 1
        print("FAIL", xs, cout, ks, stride, pad, mode, "errs (y, dx, dw, db)", ["%.2e" % e for e in errs])
print(f"{a.cases} cases, {bad} failures ({'bf16' if a.bf16 else 'fp32'} mode, seed {a.seed})")
sys.exit(1 if bad else 0)
