"""Timing of the dense GATv2 attention kernels (csrc/gnn.hip) over node / head / channel counts.  python tools/gat_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, H, C in ((24, 4, 512), (6, 4, 512), (48, 4, 512), (24, 1, 512), (24, 4, 64), (24, 4, 1024)):
    xl = torch.randn(N, H, C, device=dev, requires_grad=True)
    xr = torch.randn(N, H, C, device=dev, requires_grad=True)
    att = torch.randn(H, C, device=dev, requires_grad=True)
    A = (torch.rand(N, N, device=dev) < 0.4).float()
    g = torch.randn(N, C, device=dev)
    with torch.no_grad():
        t_f = timeit(lambda: ops.gatv2_dense(xl, xr, att, A))
    out = ops.gatv2_dense(xl, xr, att, A)
    t_b = timeit(lambda: torch.autograd.grad(out, (xl, xr, att), g, retain_graph=True))
    print(f"N={N:3d} H={H} C={C:5d}: forward {t_f:7.1f} us   backward (2 launches) {t_b:7.1f} us", flush=True)
