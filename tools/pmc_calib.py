"""Known-byte kernels for calibrating FETCH_SIZE / WRITE_SIZE on this box (MI355X guide: FETCH_SIZE reads 1/2 for wide
coalesced reads; other widths are uncalibrated).  Run under rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
dev = "cuda:0"
x = torch.randn(64, 1024, 1024, device=dev)                  # 268 MB: a 16-byte-per-lane streaming copy (torch)
for _ in range(3):
    y = x.clone()
img = torch.randn(40, 64, 64, 128, device=dev)               # 83.9 MB image, dword-per-lane gathers (flow_warp_fwd)
flow = torch.zeros(40, 2, 64, 128, device=dev)
for _ in range(3):
    z = ops.flow_warp(img, flow)
torch.cuda.synchronize()
print("calib: copy bytes", x.numel() * 4, "warp image bytes", img.numel() * 4, "flow bytes", flow.numel() * 4)
