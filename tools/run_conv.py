"""Runs one conv layer forward + backward a few times (for profiling a single kernel shape).
usage: python tools/run_conv.py N Cin H W Cout k stride pad mode [iters] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
a = sys.argv[1:]
N, Cin, H, W, Cout, k, stride, pad = (int(v) for v in a[:8])
mode = a[8]
iters = int(a[9]) if len(a) > 9 else 10
T = int(a[10]) if len(a) > 10 else 0
torch.manual_seed(0)
if T:
    x = torch.randn(N, Cin, T, H, W, device="cuda:0", requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, k, device="cuda:0") / (Cin * k ** 3) ** 0.5).requires_grad_(True)
else:
    x = torch.randn(N, Cin, H, W, device="cuda:0", requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, device="cuda:0") / (Cin * k * k) ** 0.5).requires_grad_(True)
y = ops.conv(x, w, None, stride=stride, padding=pad, padding_mode=mode)
go = torch.randn_like(y)
with ops.ConvProfiler() as prof:
    for _ in range(iters):
        y = ops.conv(x, w, None, stride=stride, padding=pad, padding_mode=mode)
        y.backward(go)
        x.grad = None; w.grad = None
for kname, v in prof.summary().items():
    print(f"{kname}: {v['ms'] / v['launches'] * 1000:.1f} us/launch  {v['flops'] / v['ms'] / 1e9:.1f} TF/s")
