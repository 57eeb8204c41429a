"""Bit-repeatability of conv forward / backward under dirty allocator state (uninitialised-read / stream-race hunt):
every CONV_CASE twice in one process, the caching allocator's free blocks poisoned with NaN in between.
    python tools/repeat_check.py [bf16]
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from c2m_amd import ops
from test_gpu_ops import CONV_CASES, WINO_CASES

bf16 = "bf16" in sys.argv
ops.set_conv_precision("bf16" if bf16 else "fp32")
dev = "cuda:0"


def poison():
    junk = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(8)]       # 2 GiB of NaN through the main pool
    s = ops._side_stream(torch.device(dev))
    with torch.cuda.stream(s):
        junk2 = [torch.full((16 << 20,), float("nan"), device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    del junk, junk2


def run(case, act):
    xs, cout, k, stride, pad, mode = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()) % 10000)
    x = torch.randn(*xs, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(cout, xs[1], *k, generator=g) / (xs[1] * 9) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(cout, generator=g).to(dev).requires_grad_(True)
    y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode, act=act)
    go = torch.randn(*y.shape, generator=g).to(dev).to(y.dtype)
    (y.float() * go.float()).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()


BIG = [   # layer shapes of the bench configurations (kernels long enough to overlap across streams)
    ((40, 256, 16, 32), 256, (3, 3), 1, 1, "reflect"),
    ((40, 128, 32, 64), 128, (3, 3), 1, 1, "reflect"),
    ((40, 64, 64, 128), 64, (3, 3), 1, 1, "reflect"),
    ((40, 128, 64, 128), 64, (3, 3), 1, 1, "reflect"),
    ((8, 32, 128, 256), 32, (3, 3), 1, 1, "reflect"),
    ((40, 64, 64, 128), 128, (4, 4), 2, 1, "reflect"),
    ((40, 128, 32, 64), 256, (4, 4), 2, 1, "reflect"),
    ((40, 32, 128, 256), 64, (4, 4), 2, 1, "reflect"),
    ((8, 32, 5, 64, 128), 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect"),
    ((8, 34, 5, 128, 256), 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "reflect"),
    ((40, 6, 128, 256), 64, (4, 4), 2, 1, "zeros"),
    ((40, 32, 128, 256), 3, (7, 7), 1, 3, "reflect"),
    ((40, 512, 8, 16), 512, (3, 3), 1, 1, "reflect"),
    ((40, 256, 16, 32), 512, (4, 4), 2, 1, "reflect"),
]
bad = 0
cases = BIG if "big" in sys.argv else CONV_CASES + WINO_CASES
for case in cases:
    for act in (None, "lrelu"):
        a = run(case, act)
        poison()
        b = run(case, act)
        names = ("y", "gx", "gw", "gb")
        diff = [n for n, p, q in zip(names, a, b) if not torch.equal(p, q)]
        nan = [n for n, p in zip(names, b) if not bool(torch.isfinite(p.float()).all())]
        if diff or nan:
            bad += 1
            print("DIFF", case, act, diff, "nan:", nan)
            for n, p_, q_ in zip(names, a, b):
                if n in diff:
                    d = (p_.float() - q_.float()).abs()
                    idx = d.flatten().nonzero().flatten()
                    print("   ", n, "differing", idx.numel(), "of", d.numel(), "max", float(d.max()), "scale", float(p_.float().abs().max()),
                          "first idx", idx[:12].tolist())
print("cases with differences:", bad)
