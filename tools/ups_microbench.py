"""Times the 2x bilinear upsample forward / backward kernels on the shapes of the bench step.
usage: python tools/ups_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops

for shape in [(40, 32, 64, 128), (40, 64, 32, 64), (40, 128, 16, 32), (40, 256, 8, 16), (8, 512, 4, 8)]:
    x = torch.randn(*shape, device="cuda:0", requires_grad=True)
    y = ops.upsample2x(x)
    go = torch.randn_like(y)
    for which in ("fwd", "bwd"):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(12):
            if it == 2:
                ev[0].record()
            if which == "fwd":
                with torch.no_grad():
                    ops.upsample2x(x)
            else:
                y = ops.upsample2x(x)
                y.backward(go)
                x.grad = None
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 100.0
        nbytes = 4 * x.numel() * 5
        print(shape, which, f"{us:.1f} us/iter", f"{nbytes / us / 1e6:.2f} TB/s (one pass of in + out)" if which == "fwd" else "")
