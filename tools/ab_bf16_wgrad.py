"""A/B of the bf16 weight-gradient kernels on one box: 16-byte loads (conv_wgrad_wide_bf16_kernel) vs the per-pixel 2-byte
gathers (C2M_WGRAD_NARROW=1), bf16 tensors.  usage: python tools/ab_bf16_wgrad.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(40, 256, 16, 32, 256, "reflect", 2), (40, 128, 64, 128, 128, "reflect", 2), (40, 32, 128, 256, 32, "reflect", 2),
          (40, 64, 64, 128, 128, "reflect", 2), (40, 128, 32, 64, 128, "zeros", 2), (8, 34, 128, 256, 32, "reflect", 3),
          (40, 64, 128, 256, 32, "reflect", 2), (40, 256, 32, 64, 128, "reflect", 2)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from c2m_amd import ops
    ops.set_conv_precision("bf16")
    for (N, Cin, H, W, Cout, mode, nd) in SHAPES:
        xs = (N, Cin, H, W) if nd == 2 else (N, Cin, 5, H, W)
        x = torch.randn(*xs, device="cuda:0").bfloat16()
        w = (torch.randn(Cout, Cin, *([3] * nd), device="cuda:0") / (Cin * 3 ** nd) ** 0.5).requires_grad_(True)
        y = ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
        go = torch.randn_like(y)
        with ops.ConvProfiler() as prof:
            for _ in range(30):
                y = ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
                y.backward(go)
                w.grad = None
        for tag, n, ms, tf in prof.table():
            if tag[1] == "wgrad":
                print((N, Cin, H, W, Cout, mode, nd), f"{ms / n * 1000:.1f} us  {tf:.1f} TF/s", flush=True)
else:
    for narrow in ("", "1"):
        env = dict(os.environ)
        if narrow:
            env["C2M_WGRAD_NARROW"] = "1"
        print("==== C2M_WGRAD_NARROW =", narrow or "0", flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
