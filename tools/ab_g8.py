"""bf16 layers outside the NC8 patch forms: the NCHW 2-byte-gather kernel (C2M_G8=0 rule) against the NC8 gather form
(conv_gather_nc8_kernel) on bench shapes of configs[2-4]; forward and data gradient per tile variant, results compared.
    python tools/ab_g8.py [iters]        AB_VARIANTS=0,2,5 picks the variants (0 = the rule)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEV = "cuda:0"
SHAPES = [  # x shape, Cout, kernel, stride, padding, mode     (configs[3]: 40 folded frames / 8 clips at 128x256)
    ((40, 128, 32, 64), 256, (4, 4), 2, 1, "reflect"),          # data gradient over 17 x 33 class planes
    ((40, 256, 16, 32), 512, (4, 4), 2, 1, "reflect"),
    ((40, 512, 8, 16), 512, (4, 4), 2, 1, "reflect"),
    ((40, 512, 8, 16), 512, (3, 3), 1, 1, "zeros"),             # 3x3 on 8 x 16 maps
    ((8, 45, 4, 128, 256), 32, (4, 4, 4), 2, 1, "reflect"),     # motion encoder head
    ((8, 32, 4, 64, 128), 64, (3, 4, 4), (1, 2, 2), 1, "reflect"),
    ((8, 64, 4, 32, 64), 128, (3, 4, 4), (1, 2, 2), 1, "reflect"),
    ((8, 128, 4, 16, 32), 256, (3, 4, 4), (1, 2, 2), 1, "reflect"),
    ((8, 512, 3, 8, 16), 512, (3, 3, 3), 1, 1, "reflect"),
    ((40, 256, 32, 64), 64, (1, 1), 1, 0, "zeros"),
    ((20, 128, 64, 128), 256, (4, 4), 2, 1, "reflect"),         # configs[2]
]
ops.set_conv_precision("bf16")


def run(g8, xs, Cout, k, stride, pad, mode):
    ops._G8 = g8
    ops._geom_cache.clear()
    g = torch.Generator().manual_seed(xs[1] * 7 + xs[-1])
    x = torch.randn(*xs, generator=g).to(DEV).bfloat16().requires_grad_(True)
    taps = 1
    for v in k:
        taps *= v
    w = (torch.randn(Cout, xs[1], *k, generator=g) / (xs[1] * taps) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode, act="lrelu")
    go = torch.randn(*y.shape, generator=g).to(DEV).bfloat16()
    y.backward(go)
    with ops.ConvProfiler() as prof:
        for _ in range(iters):
            x.grad = None
            if os.environ.get("AB_LAYOUT", "1") != "0":
                x.__dict__.pop("_c2m_nc8", None)      # time the layout pass of the input as well (the bench step pays it once per tensor)
            y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode, act="lrelu")
            y.backward(go)
    tab = {}
    for r in prof.table():
        if r[0][0] == "igemm_bf16":
            e = tab.setdefault(r[0][1], [0.0, 0.0])
            tab["S" + r[0][1]] = r[0][-2] if r[0][-1] == "nc8g" else r[0][-1]
            e[0] += r[2] * 1000 / iters
            e[1] += r[2] * r[3]
    return y.detach(), x.grad.clone(), {k_: ((v[0], v[1] / (v[0] * iters / 1000)) if isinstance(v, list) else v) for k_, v in tab.items()}


VARIANTS = [int(v) for v in os.environ.get("AB_VARIANTS", "0").split(",")]
for shp in SHAPES:
    y0, g0, t0 = run(False, *shp)
    msg = f"{str(shp):62s} old fwd {t0['fwd'][0]:6.1f} us {t0['fwd'][1]:4.0f} TF/s (S {t0['Sfwd']}) dgrad {t0['dgrad'][0]:6.1f} us {t0['dgrad'][1]:4.0f} (S {t0['Sdgrad']})"
    for v in VARIANTS:
        ops._G8_VARIANT = v
        y1, g1, t1 = run(True, *shp)
        ey = float((y1.float() - y0.float()).abs().max() / y0.float().abs().max())
        eg = float((g1.float() - g0.float()).abs().max() / g0.float().abs().max())
        msg += f" | v{v}: fwd {t1['fwd'][0]:6.1f} us {t1['fwd'][1]:4.0f} (S {t1['Sfwd']}) dgrad {t1['dgrad'][0]:6.1f} us {t1['dgrad'][1]:4.0f} (S {t1['Sdgrad']}) diff {ey:.0e} {eg:.0e}"
    print(msg, flush=True)
