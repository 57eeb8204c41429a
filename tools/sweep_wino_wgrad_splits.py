"""Sweep the region-split count of the Winograd weight-gradient launch (C2M_WINO_WG_SPLITS tuning hook) on the bench's
eligible layers.  usage: python tools/sweep_wino_wgrad_splits.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(40, 256, 16, 32, 256), (40, 128, 64, 128, 128), (40, 64, 64, 128, 128), (40, 128, 32, 64, 128),
          (40, 128, 32, 64, 256), (40, 256, 64, 128, 64), (40, 256, 16, 32, 128), (40, 128, 16, 32, 512),
          (40, 128, 64, 128, 64), (40, 256, 32, 64, 128), (40, 1536, 8, 16, 256), (40, 768, 16, 32, 128),
          (40, 384, 32, 64, 64), (40, 64, 32, 64, 64), (40, 192, 64, 128, 32), (40, 512, 16, 32, 256),
          (40, 256, 8, 16, 512)]
ops._WINO_WGRAD = "force"
L = _lib.lib()
for N, Cin, H, W, Cout in SHAPES:
    regions = N * (H // 8) * (W // 16)
    tiles = -(-Cout // 64) * -(-Cin // 32)
    os.environ.pop("C2M_WINO_WG_SPLITS", None)
    default = L.c2m_wino_wgrad_splits(Cout, Cin, N, H, W)
    cands = sorted({default} | {max(1, min(regions, t // tiles)) for t in (256, 512, 768, 1024, 1536, 2048)}
                   | {max(1, min(regions, regions // p)) for p in (4, 5, 8, 10, 16, 20, 32, 40)})
    out = []
    torch.manual_seed(0)
    x = torch.randn(N, Cin, H, W, device="cuda:0")
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5).requires_grad_(True)
    for S in cands:
        if S * 16 * Cin * Cout * 4 > 1 << 30:
            continue
        os.environ["C2M_WINO_WG_SPLITS"] = str(S)
        ops._geom_cache.clear()
        y = ops.conv(x, w, None, stride=1, padding=1, padding_mode="reflect")
        go = torch.ones_like(y)
        y.backward(go); w.grad = None
        with ops.ConvProfiler() as prof:
            for _ in range(iters):
                y = ops.conv(x, w, None, stride=1, padding=1, padding_mode="reflect")
                y.backward(go); w.grad = None
        s = prof.summary()["wino_wgrad"]
        Seff = L.c2m_wino_wgrad_splits(Cout, Cin, N, H, W)
        out.append((Seff, Seff * tiles, -(-regions // Seff), s["ms"] / s["launches"] * 1000))
    best = min(out, key=lambda t: t[3])
    print(f"Cin {Cin} Cout {Cout} {H}x{W}: regions {regions} tiles {tiles} default S {default} | best S {best[0]} "
          f"({best[1]} wgs, {best[2]} regions each) {best[3]:.1f} us | " +
          " ".join(f"S{a}/{b}w/{c}r:{d:.0f}{'*' if a == default else ''}" for a, b, c, d in out), flush=True)
