"""bf16 3x3 stride-1 layers: the NCHW 2-byte-gather kernels of round 3 (C2M_NC8=0 rules) against the channel-blocked kernel
(conv_nc8.hip) on bench shapes of configs[2-4]; forward and data gradient, bit-comparison included.
    python tools/ab_nc8.py [iters]        (one process: the plan cache is cleared between the two modes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops, _lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEV = "cuda:0"
SHAPES = [  # N, Cin, H, W, Cout, padding mode          (configs[3]: 40 folded frames at 128x256)
    (40, 64, 128, 256, 64, "zeros"), (40, 128, 64, 128, 128, "zeros"), (40, 256, 32, 64, 256, "zeros"), (40, 512, 16, 32, 512, "zeros"),
    (40, 128, 64, 128, 128, "reflect"), (40, 64, 64, 128, 128, "reflect"), (40, 256, 32, 64, 128, "reflect"),
    (40, 32, 128, 256, 32, "reflect"), (40, 256, 16, 32, 256, "reflect"), (8, 64, 128, 256, 32, "reflect"),
    (20, 256, 64, 128, 256, "zeros"), (20, 64, 256, 512, 64, "zeros"),      # configs[2] (256x512)
]
ops.set_conv_precision("bf16")
L = _lib.lib()


def run(mode_nc8, N, Cin, H, W, Cout, pad):
    ops._NC8 = mode_nc8
    ops._geom_cache.clear()
    g = torch.Generator().manual_seed(Cin * 7 + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV).bfloat16().requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(DEV)
    y = ops.conv(x, w, b, stride=1, padding=1, padding_mode=pad, act="lrelu")
    go = torch.randn(N, Cout, H, W, generator=g).to(DEV).bfloat16()
    y.backward(go)
    with ops.ConvProfiler() as prof:
        for _ in range(iters):
            x.grad = w.grad = None
            y = ops.conv(x, w, b, stride=1, padding=1, padding_mode=pad, act="lrelu")
            y.backward(go)
    tab = {r[0][1]: (r[2] / r[1] * 1000, r[3], r[0][-1]) for r in prof.table() if r[0][0] in ("igemm_bf16", "wgrad_bf16")}
    return y.detach(), x.grad.clone(), tab, w.grad.clone()


VARIANTS = [int(v) for v in os.environ.get("AB_VARIANTS", "0").split(",")]
for shp in SHAPES:
    y0, g0, t0, w0 = run(False, *shp)
    msg = f"{str(shp):40s} old fwd {t0['fwd'][0]:6.1f} us {t0['fwd'][1]:4.0f} TF/s dgrad {t0['dgrad'][0]:6.1f} us {t0['dgrad'][1]:4.0f} wgrad {t0['wgrad'][0]:6.1f} us {t0['wgrad'][1]:4.0f}"
    for v in VARIANTS:
        ops._NC8_VARIANT = v if not (v in (4, 5) and shp[4] > 32) and not (shp[4] <= 32 and v in (1, 2, 3, 6)) else (5 if v in (2, 6) else 4)
        y1, g1, t1, w1 = run(True, *shp)
        same = torch.equal(y0, y1), torch.equal(g0, g1)
        werr = float((w1 - w0).abs().max() / w0.abs().max())
        msg += (f" | v{ops._NC8_VARIANT}: fwd {t1['fwd'][0]:6.1f} us {t1['fwd'][1]:4.0f} dgrad {t1['dgrad'][0]:6.1f} us {t1['dgrad'][1]:4.0f} {t1['dgrad'][2]} "
                f"wgrad {t1['wgrad'][0]:6.1f} us {t1['wgrad'][1]:4.0f} {t1['wgrad'][2]} same={int(same[0])}{int(same[1])} gw rel diff {werr:.1e}")
    print(msg, flush=True)
# the layout pass alone
# 4x4 stride-2 forward: gather kernel vs the parity-plane NC8 kernel (+ fp32 reference error)
for N, Cin, H, W, Cout, pad in ((40, 64, 64, 128, 128, "reflect"), (40, 128, 32, 64, 256, "reflect"), (40, 32, 128, 256, 64, "reflect"),
                                (40, 256, 16, 32, 512, "reflect"), (8, 64, 128, 256, 128, "zeros"), (40, 15, 128, 256, 64, "reflect")):
    res = {}
    for mode in (False, True):
        ops._NC8_S2 = mode
        ops._geom_cache.clear()
        g = torch.Generator().manual_seed(Cin + H)
        x = torch.randn(N, Cin, H, W, generator=g).to(DEV).bfloat16()
        w = (torch.randn(Cout, Cin, 4, 4, generator=g) / (Cin * 16) ** 0.5).to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        with torch.no_grad():
            y = ops.conv(x, w, b, stride=2, padding=1, padding_mode=pad, act="lrelu")
            with ops.ConvProfiler() as prof:
                for _ in range(iters):
                    ops.conv(x, w, b, stride=2, padding=1, padding_mode=pad, act="lrelu")
            r = [r for r in prof.table() if r[0][1] == "fwd"][0]
        res[mode] = (y.float(), r[2] / r[1] * 1000, r[3], r[0][-1])
    xp = torch.nn.functional.pad(x.float(), (1, 1, 1, 1), mode="reflect" if pad == "reflect" else "constant")
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xp, w.bfloat16().float(), b, stride=2), 0.2)
    e0, e1 = (float((res[m][0] - ref).abs().max() / ref.abs().max()) for m in (False, True))
    print(f"s2 {(N, Cin, H, W, Cout, pad)}: gather {res[False][1]:6.1f} us {res[False][2]:4.0f} TF/s -> nc8 {res[True][1]:6.1f} us {res[True][2]:4.0f} TF/s "
          f"({res[True][3]}) | max err vs fp32 conv of the same bf16 operands: gather {e0:.1e} nc8 {e1:.1e}", flush=True)
for N, C, H, W in ((40, 128, 64, 128), (40, 64, 128, 256), (40, 256, 32, 64)):
    x = torch.randn(N, C, H, W, device=DEV).bfloat16()
    ref = x.view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    got = ops._to_nc8(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        x.__dict__.pop("_c2m_nc8", None)              # _to_nc8 caches its result on the tensor: time the kernel, not the cache
        ops._to_nc8(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    print(f"nchw_to_nc8 {(N, C, H, W)}: equal {torch.equal(ref, got)}  {us:.1f} us  {2 * x.numel() * 2 / us / 1e6:.2f} TB/s", flush=True)
