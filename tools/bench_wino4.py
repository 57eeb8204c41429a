"""F(4x4,3x3) kernel (conv_wino4.hip) against the shipped F(2x2,3x3) kernel on bench-model layer
shapes: forward launches through the two C ABIs, torch events on the launch stream.   python tools/bench_wino4.py [iters]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from c2m_amd import _lib
L2 = _lib.lib()
L4 = ctypes.CDLL(_lib.LIB_PATH)
L4.c2m_wino4_upack_floats.restype = ctypes.c_long
L4.c2m_wino4_upack_floats.argtypes = [ctypes.c_int, ctypes.c_int]
L4.c2m_wino4_filter_transform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L4.c2m_conv_wino4.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = lambda t: ctypes.c_void_p(t.data_ptr())
if os.environ.get("W4_SHAPES"):                # "N,Cin,H,W,Cout,reflect;..." instead of the bench-model list
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["W4_SHAPES"].split(";")]
else:
  SHAPES = [(40, 512, 16, 32, 512, 0), (40, 256, 32, 64, 256, 0), (40, 256, 16, 32, 256, 1), (40, 128, 64, 128, 128, 1),
          (40, 128, 64, 128, 128, 0), (40, 64, 128, 256, 64, 0), (40, 128, 32, 64, 128, 1), (40, 256, 32, 64, 128, 1)]
print("N Cin H W Cout reflect | F(2x2) us  TF/s | F(4x4) us  TF/s | speed-up | max rel err F(4x4)")
for (N, Cin, H, W, Cout, refl) in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(N, Cin, H, W, device="cuda:0"); w = torch.randn(Cout, Cin, 3, 3, device="cuda:0") / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device="cuda:0") * 0.1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = np.zeros(_lib.WINO_GEOM.LEN, dtype=np.int64)      # include/c2m_geom.h: C2M_WG_LEN (a shorter block would hand the kernel a garbage ring pointer)
    g[:18] = [Cout, Cin, N, H, W, H, W, -1, -1, refl, Cin * H * W, H * W, W, Cout * H * W, H * W, W, 0, 4 * N * Cin * H * W]
    gp = g.ctypes.data_as(ctypes.c_void_p)
    U2 = torch.empty(L2.c2m_wino_upack_floats(Cout, Cin), device="cuda:0")
    U4 = torch.empty(L4.c2m_wino4_upack_floats(Cout, Cin), device="cuda:0")
    assert L2.c2m_wino_filter_transform(p(w), p(U2), Cout, Cin, 0, st) == 0
    assert L4.c2m_wino4_filter_transform(p(w), p(U4), Cout, Cin, 0, st) == 0
    y2 = torch.empty(N, Cout, H, W, device="cuda:0"); y4 = torch.empty_like(y2)
    f2 = lambda: L2.c2m_conv_wino(p(U2), p(x), p(y2), None, p(b), gp, 0, 0.0, st)
    f4 = lambda: L4.c2m_conv_wino4(p(U4), p(x), p(y4), None, p(b), gp, 0, 0.0, st)
    res = []
    for f in (f2, f4):
        for _ in range(3):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    fl = 2.0 * Cout * Cin * 9 * N * H * W
    xp = F.pad(x[:1].double(), (1, 1, 1, 1), mode="reflect") if refl else F.pad(x[:1].double(), (1, 1, 1, 1))
    ref = F.conv2d(xp, w.double(), b.double())
    err4 = float((y4[:1].double() - ref).abs().max() / ref.abs().max())
    err2 = float((y2[:1].double() - ref).abs().max() / ref.abs().max())
    print(f"{N} {Cin} {H} {W} {Cout} {refl} | {res[0]:8.1f} {fl / res[0] / 1e6:6.1f} | {res[1]:8.1f} {fl / res[1] / 1e6:6.1f} | {res[0] / res[1]:5.2f}x | {err4:.2e} (F(2x2): {err2:.2e})", flush=True)
