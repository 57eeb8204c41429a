"""Phase timing inside conv_wino4_kernel (workgroup 0): needs the tuning build  python -m c2m_amd.build w4trace -DW4_TRACE  and
C2M_AMD_LIB=c2m_amd/lib/libc2m_hip_w4trace.so.   python tools/trace_wino4.py N Cin H W Cout"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c2m_amd import _lib
L = _lib.lib()
N, Cin, H, W, Cout = (int(v) for v in sys.argv[1:6])
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.randn(N, Cin, H, W, device="cuda:0"); w = torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5
b = torch.zeros(Cout, device="cuda:0"); y = torch.empty(N, Cout, H, W, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = np.zeros(_lib.WINO_GEOM.LEN, dtype=np.int64)      # include/c2m_geom.h: C2M_WG_LEN (a shorter block would hand the kernel a garbage ring pointer)
g[:18] = [Cout, Cin, N, H, W, H, W, -1, -1, 0, Cin * H * W, H * W, W, Cout * H * W, H * W, W, 0, 4 * N * Cin * H * W]
L.c2m_wino4_upack_floats.restype = ctypes.c_long
U = torch.empty(L.c2m_wino4_upack_floats(Cout, Cin), device="cuda:0")
L.c2m_wino4_filter_transform(p(w), p(U), Cout, Cin, 0, st)
for _ in range(3):
    L.c2m_conv_wino4(p(U), p(x), p(y), None, p(b), g.ctypes.data_as(ctypes.c_void_p), 0, ctypes.c_float(0.0), st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16 * 4 + 16))()
L.c2m_wino4_trace_read.argtypes = [ctypes.c_void_p]
assert L.c2m_wino4_trace_read(buf) == 0
a = np.array(buf[:8 * 16 * 4], dtype=np.int64).reshape(8, 16, 4)
tb = np.array(buf[8 * 16 * 4:], dtype=np.int64).reshape(8, 2)
nint = min(16, (Cin + 7) // 8)
t0 = tb[:, 0].min()
print("s_memtime ticks (100 MHz on gfx950? compare with the kernel's total): kernel total per wave:", (tb[:, 1] - tb[:, 0]).tolist())
print("wave | t_first | per interval (mean over intervals 1..): barrier->mma_start, mma, mma_end->end, end->next barrier, interval")
for wv in range(8):
    tf = ((wv ^ (wv >> 2)) & 1)
    s = a[wv, 1:nint]
    if len(s) < 2:
        continue
    pre, mm, post = (s[:, 1] - s[:, 0]).mean(), (s[:, 2] - s[:, 1]).mean(), (s[:, 3] - s[:, 2]).mean()
    nxt = (s[1:, 0] - s[:-1, 3]).mean(); itv = (s[1:, 0] - s[:-1, 0]).mean()
    print(f"{wv} | {tf} | {pre:8.1f} {mm:8.1f} {post:8.1f} {nxt:8.1f} {itv:8.1f}")
print("prologue (kernel begin -> first barrier of the loop):", (a[:, 0, 0] - tb[:, 0]).tolist())
print("epilogue (end of last traced interval -> kernel end; only exact when nchunks <= 16):", (tb[:, 1] - a[:, nint - 1, 3]).tolist())
