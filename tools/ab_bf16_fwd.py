"""bf16 forward conv timing on a few bench shapes (configs[2]).  usage: python tools/ab_bf16_fwd.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [(20, 256, 64, 128, 256, "zeros"), (20, 512, 32, 64, 512, "zeros"), (20, 128, 128, 256, 128, "reflect"),
          (20, 64, 128, 256, 128, "reflect"), (20, 64, 256, 512, 64, "zeros"), (20, 256, 32, 64, 256, "reflect")]
ops.set_conv_precision("bf16")
for N, Cin, H, W, Cout, mode in SHAPES:
    x = torch.randn(N, Cin, H, W, device="cuda:0")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5
    with torch.no_grad():
        ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
        with ops.ConvProfiler() as prof:
            for _ in range(iters):
                ops.conv(x, w, None, stride=1, padding=1, padding_mode=mode)
        s = prof.summary()["igemm_bf16"]
    print(f"{(N, Cin, H, W, Cout, mode)}: {s['ms'] / s['launches'] * 1000:.1f} us  {s['flops'] / s['ms'] / 1e9:.0f} TF/s", flush=True)
