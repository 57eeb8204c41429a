"""Micro-benchmark of one conv shape through the C ABI (fwd, dgrad, wgrad), for rocprofv3 counter runs.
usage: python tools/conv_microbench.py N Cin H W Cout k stride pad mode [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops

N, Cin, H, W, Cout, k, stride, pad = (int(a) for a in sys.argv[1:9])
mode = sys.argv[9] if len(sys.argv) > 9 else "zeros"
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 20
which = sys.argv[11] if len(sys.argv) > 11 else "all"
dev = "cuda:0"
if os.environ.get("C2M_BENCH_BF16"):          # bf16 data path (tools/pmc_gather_bf16.sh)
    ops.set_conv_precision("bf16")
x = torch.randn(N, Cin, H, W, device=dev, requires_grad=True)
w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).requires_grad_(True)
b = torch.zeros(Cout, device=dev, requires_grad=True)
y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode)
go = torch.randn_like(y)
with ops.ConvProfiler() as prof:
    for _ in range(iters):
        if which == "fwd":
            with torch.no_grad():
                ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode)
        else:
            y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode)
            y.backward(go)
            x.grad = w.grad = b.grad = None
for tag, n, ms, tf in prof.table():
    print(tag, f"{ms / n * 1000:.1f} us/launch  {tf:.1f} TFLOP/s")
