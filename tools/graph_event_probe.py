"""Probe (GPU box): can HIP events be recorded INSIDE a HIP-graph replay and timed afterwards?  hipEventRecordWithFlags(...,
hipEventRecordExternal) during a stream capture makes an event-record NODE; after each replay hipEventElapsedTime between two such
events should give the GPU time of the kernels between them.  Prints the elapsed time of a known kernel sequence per replay next to
the same sequence timed eagerly.  (Decides whether bench.py's roofline events can live inside replays.)"""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableSystemFence = 0x20000000


def ev_create():
    h = ctypes.c_void_p()
    rc = hip.hipEventCreateWithFlags(ctypes.byref(h), ctypes.c_uint(hipEventDisableSystemFence))
    assert rc == 0, rc
    return h


def record(e, stream, flags):
    rc = hip.hipEventRecordWithFlags(e, ctypes.c_void_p(stream), ctypes.c_uint(flags))
    return rc


def elapsed(a, b):
    ms = ctypes.c_float()
    rc = hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
    return rc, ms.value


dev = torch.device("cuda", 0)
x = torch.randn(8192, 8192, device=dev)
y = torch.empty_like(x)


def work(n):
    for _ in range(n):
        torch.mul(x, 1.0001, out=y)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    work(3)
    torch.cuda.synchronize()
    raw = side.cuda_stream
    e = [ev_create() for _ in range(4)]
    # eager reference
    record(e[0], raw, 0); work(4); record(e[1], raw, 0); work(8); record(e[2], raw, 0)
    torch.cuda.synchronize()
    print("eager: 4 kernels", elapsed(e[0], e[1]), "8 kernels", elapsed(e[1], e[2]))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        rcs = [record(e[0], raw, 1)]
        work(4)
        rcs.append(record(e[1], raw, 1))
        work(8)
        rcs.append(record(e[2], raw, 1))
    print("record rcs inside capture:", rcs)
    for it in range(3):
        g.replay()
        torch.cuda.synchronize()
        print(f"replay {it}: 4 kernels", elapsed(e[0], e[1]), "8 kernels", elapsed(e[1], e[2]))
