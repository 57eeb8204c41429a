"""A/B of the weight-gradient kernels on the bench's 3x3 stride-1 layers: direct (conv_wgrad_kernel) vs Winograd
(conv_wino_wgrad_kernel).  usage: python tools/ab_wino_wgrad.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# (N, Cin, H, W, Cout, mode) of the configs[1] step
SHAPES = [(40, 256, 16, 32, 256, "reflect"), (40, 128, 64, 128, 128, "reflect"), (40, 64, 64, 128, 128, "reflect"),
          (40, 128, 32, 64, 128, "reflect"), (40, 128, 32, 64, 256, "reflect"), (40, 256, 64, 128, 64, "reflect"),
          (40, 256, 16, 32, 128, "reflect"), (40, 128, 16, 32, 512, "reflect"), (40, 128, 64, 128, 64, "reflect"),
          (40, 256, 32, 64, 128, "reflect"), (40, 1536, 8, 16, 256, "reflect"), (40, 768, 16, 32, 128, "reflect"),
          (40, 64, 32, 64, 64, "reflect"), (40, 384, 32, 64, 64, "reflect"), (40, 192, 64, 128, 32, "reflect")]
if os.environ.get("AB_EXTRA"):      # layers the auto rule currently leaves on the direct kernel
    SHAPES = [(40, 32, 128, 256, 32, "reflect"), (40, 64, 128, 256, 32, "reflect"), (40, 32, 128, 256, 64, "reflect"),
              (40, 96, 128, 256, 32, "reflect"), (40, 64, 64, 128, 64, "reflect"), (40, 128, 128, 256, 32, "reflect")]
for shape in SHAPES[:int(os.environ.get('AB_SHAPES', len(SHAPES)))]:
    N, Cin, H, W, Cout, mode = shape
    res = {}
    for which in ("off", "force"):
        ops._WINO_WGRAD = which
        ops._geom_cache.clear()
        torch.manual_seed(0)
        x = torch.randn(N, Cin, H, W, device="cuda:0")
        w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5).requires_grad_(True)
        y = ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
        go = torch.randn_like(y)
        y.backward(go)
        ref = w.grad.clone()
        w.grad = None
        with ops.ConvProfiler() as prof:
            for _ in range(iters):
                y = ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
                y.backward(go)
                w.grad = None
        s = prof.summary()
        k = "wino_wgrad" if which == "force" and "wino_wgrad" in s else "wgrad"
        res[which] = (k, s[k]["ms"] / s[k]["launches"] * 1000, s[k]["flops"] / s[k]["ms"] / 1e9, ref)
    err = float((res["off"][3] - res["force"][3]).abs().max() / res["off"][3].abs().max())
    print(f"{shape}: direct {res['off'][1]:.1f} us {res['off'][2]:.1f} TF/s | {res['force'][0]} {res['force'][1]:.1f} us "
          f"{res['force'][2]:.1f} TF/s | speedup {res['off'][1] / res['force'][1]:.2f} | rel diff {err:.1e}", flush=True)
