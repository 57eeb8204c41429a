"""A/B on one box: data gradient of the bench model's reflect-padded 3x3 layers, padded-domain route (two-target epilogue + fold,
rounds 1-4) vs interior + ring route (round 5, conv_ring.hip), per shape, HIP-event timed.  python tools/bench_ring.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops

SHAPES = [  # (N, Cin, H, W, Cout): x shape + output channels (the table of profiles/r04_conv_shape_table.txt, reflect dgrads)
    (40, 128, 64, 128, 128), (40, 64, 64, 128, 128), (40, 256, 64, 128, 64), (40, 128, 64, 128, 64), (40, 192, 64, 128, 32),
    (40, 128, 32, 64, 128), (40, 128, 32, 64, 256), (40, 256, 32, 64, 128), (40, 384, 32, 64, 64),
    (40, 256, 16, 32, 256), (40, 768, 16, 32, 128), (40, 32, 128, 256, 32), (40, 64, 128, 256, 32),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)


def run(mode, shape, min_pix=None):
    ops._RING = mode
    if min_pix is not None:
        ops._RING_MIN_PIX = min_pix
    ops._geom_cache.clear()
    N, Cin, H, W, Cout = shape
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
    gy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    pl = ops._plan(x, w, (1, 1, 1), (0, 1, 1), True)
    for _ in range(3):
        gx = ops._conv_dgrad(pl, w, gy, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gx = ops._conv_dgrad(pl, w, gy, True)
    e1.record()
    torch.cuda.synchronize()
    total_us = e0.elapsed_time(e1) / reps * 1000.0
    if pl.ring_dgrad and ops._RING_BUFFER:  # the ring launch alone (buffer form)
        L = ops._lib.lib()
        Ar = ops._ring_pack(w, Cout, Cin)
        r_l = (max(H, W) + 3) // 4 * 4
        R = torch.empty(N * Cin * 4 * r_l, device=dev)
        e0.record()
        for _ in range(reps):
            L.c2m_reflect_ring_buffer(ops._p(Ar), ops._p(gy), ops._p(R), N, Cout, Cin, H, W, r_l, ops._stream())
        e1.record()
        torch.cuda.synchronize()
        run.ring_us = e0.elapsed_time(e1) / reps * 1000.0
    elif pl.ring_dgrad:                     # the ring launch alone
        L = ops._lib.lib()
        Ar = ops._ring_pack(w, Cout, Cin)
        e0.record()
        for _ in range(reps):
            L.c2m_reflect_ring_dgrad(ops._p(Ar), ops._p(w), ops._p(gy), ops._p(gx), N, Cout, Cin, H, W, ops._stream())
        e1.record()
        torch.cuda.synchronize()
        run.ring_us = e0.elapsed_time(e1) / reps * 1000.0
    route = ("ring+" if pl.ring_dgrad else "padded+") + ("F4" if pl.wino4_dgrad else ("F2" if pl.wino_dgrad else "direct"))
    return total_us, route, gx


print(f"{'shape (N,Cin,H,W,Cout)':32s} {'padded us':>10s} {'route':>12s} {'ring us':>10s} {'route':>10s} {'speedup':>8s} max|diff|/scale")
for sh in SHAPES:
    ta, ra, ga = run("off", sh)
    tb, rb, gb = run("auto", sh, 1)
    d = float((ga - gb).abs().max() / ga.abs().max())
    print(f"{str(sh):32s} {ta:10.1f} {ra:>12s} {tb:10.1f} {rb:>10s} {ta / tb:8.2f} {d:.2e}   ring launch alone {getattr(run, 'ring_us', 0):.1f} us")
