"""One bf16 3x3 layer, forward + backward, for rocprofv3 counter passes: python tools/run_nc8_one.py N Cin H W Cout pad variant [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
N, Cin, H, W, Cout = (int(v) for v in sys.argv[1:6])
pad, variant = sys.argv[6], int(sys.argv[7])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
ops.set_conv_precision("bf16")
ops._NC8 = variant >= 0
ops._NC8_VARIANT = max(variant, 0)
x = torch.randn(N, Cin, H, W, device="cuda:0").bfloat16().requires_grad_(True)
w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5).requires_grad_(True)
go = torch.randn(N, Cout, H, W, device="cuda:0").bfloat16()
for _ in range(iters):
    y = ops.conv(x, w, None, stride=1, padding=1, padding_mode=pad)
    y.backward(go)
    x.grad = w.grad = None
torch.cuda.synchronize()
