"""Stream-concurrency stress of the HIP kernels (VERDICT r03 "what's weak" 1 / "next" 2).

Every kernel family of the step (victims) runs on stream A while stream B runs a foreign kernel stream (aggressors):
  bf16_igemm   the bf16 gather kernel conv_igemm_kernel<..., bf16> (forward + data gradient of a 4x4 stride-2 layer) -- the
               neighbour next to which the SLP-vectorised build of conv_thin_wgrad_rows_kernel returned wrong sums in round 3
  bf16_patch   the bf16 LDS-patch kernel + the wide bf16 weight gradient (3x3 layer, forward + backward)
  fp32_wino    the fp32 Winograd kernels (hand-written v_pk_add_f32) as the neighbour of everything else
  bandwidth    a reduce-copy shaped stream: out = a + b over 3 x 256 MB (what RCCL's ring kernels look like to the memory system)
  rccl         dist.all_reduce of a 64 MB bucket on a 1-rank "nccl" group (RCCL's own kernel on its own stream)
Outputs (forward value, data / weight / bias gradients, optimizer state) must be BIT-IDENTICAL to the solo run, in both launch
orders (neighbour first; victim first with the neighbour arriving while it runs).

    python tools/concurrency_stress.py [--victims a,b] [--aggressors x,y] [--reps 3] [--list]
    C2M_AMD_LIB=c2m_amd/lib/libc2m_hip_slp.so python tools/concurrency_stress.py ...      # a tuning build
Exit code 1 if any output differed.  tests/test_gpu_concurrency.py runs the same functions under pytest.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from c2m_amd import ops

DEV = torch.device("cuda:0")


def _rnd(seed, *shape, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def _conv_case(xs, cout, k, stride, pad, mode, act="lrelu", seed=0, bias=True):
    """fn() -> [y, gx, gw, gb] of one conv layer forward + backward on the CURRENT stream (inputs are fixed tensors)."""
    x0 = _rnd(seed, *xs)
    w0 = _rnd(seed + 1, cout, xs[1], *k, scale=1.0 / (xs[1] * k[-1] * k[-2]) ** 0.5)
    b0 = _rnd(seed + 2, cout) if bias else None
    go = None

    def fn():
        nonlocal go
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True) if bias else None
        y = ops.conv(x, w, b, stride=stride, padding=pad, padding_mode=mode, act=act)
        if go is None:
            go = _rnd(seed + 3, *y.shape).to(y.dtype)
        y.backward(go)
        return [y.detach(), x.grad, w.grad] + ([b.grad] if bias else [])
    return fn


def make_victims():
    v = {}
    # direct MFMA kernels: gather forward, class-batched stride-2 data gradient, direct weight gradient
    v["direct_s2"] = _conv_case((40, 64, 64, 128), 128, (4, 4), 2, 1, "reflect", seed=10)
    v["direct_3d_s2"] = _conv_case((8, 32, 5, 64, 128), 64, (3, 4, 4), (1, 2, 2), (1, 1, 1), "reflect", seed=20)
    # LDS-patch kernel (Cin < 32: not a Winograd layer)
    v["patch3x3"] = _conv_case((8, 16, 128, 256), 32, (3, 3), 1, 1, "reflect", seed=30)
    # <= 4-output-channel vector-ALU kernels: 7x7 RGB head (thin rows forward, conv_thin_wgrad_rows_kernel), 3x3 flow head
    v["thin7x7"] = _conv_case((40, 32, 128, 256), 3, (7, 7), 1, 3, "reflect", act="sigmoid", seed=40)
    v["thin3x3"] = _conv_case((40, 32, 128, 256), 2, (3, 3), 1, 1, "reflect", act=None, seed=50)
    # Winograd F(2x2,3x3): forward, padded-domain data gradient (GEN region shape), Winograd weight gradient
    v["wino2_deep"] = _conv_case((40, 256, 16, 32), 256, (3, 3), 1, 1, "reflect", act="relu", seed=60)
    v["wino2_32rows"] = _conv_case((8, 32, 128, 256), 32, (3, 3), 1, 1, "reflect", seed=70)
    # Winograd F(4x4,3x3) forward + zero-pad data gradient (full 16x32 regions), reflect variant (F(2x2) data gradient)
    v["wino4_zeros"] = _conv_case((40, 128, 64, 128), 128, (3, 3), 1, 1, "zeros", act="relu", seed=80)
    v["wino4_reflect"] = _conv_case((40, 64, 64, 128), 64, (3, 3), 1, 1, "reflect", seed=90)
    # 3x3x3 layer as a 2-D Winograd over (time tap, channel)
    v["wino3d"] = _conv_case((8, 34, 5, 128, 256), 32, (3, 3, 3), 1, 1, "reflect", seed=100)

    # bf16 data path, channel-blocked kernels (conv_nc8.hip: counted-wait LDS-DMA, transposed LDS reads): 3x3 forward / data /
    # weight gradient and the stride-2 parity form
    def _bf16_case(*a, **k):
        fn = _conv_case(*a, **k)

        def run():
            with ops.conv_precision("bf16"):
                return fn()
        return run
    v["nc8_3x3"] = _bf16_case((40, 128, 32, 64), 128, (3, 3), 1, 1, "reflect", seed=160)
    v["nc8_s2"] = _bf16_case((40, 64, 64, 128), 128, (4, 4), 2, 1, "reflect", seed=170)

    # norm + activation: batch statistics, affine instance norm, SPADE
    xn, gam, bet = _rnd(110, 40, 128, 32, 64), _rnd(111, 128), _rnd(112, 128)
    gbm, gon = _rnd(113, 40, 256, 32, 64, scale=0.3), _rnd(114, 40, 128, 32, 64)

    def norm():
        outs = []
        for kind in ("bn", "in", "spade"):
            x = xn.clone().requires_grad_(True)
            g, b = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
            m = gbm.clone().requires_grad_(True)
            if kind == "bn":
                rm, rv = torch.zeros(128, device=DEV), torch.ones(128, device=DEV)
                y = ops.batch_norm_act(x, g, b, rm, rv, act="lrelu")
                extra = [rm, rv]
            elif kind == "in":
                y, extra = ops.instance_norm_act(x, g, b, act="lrelu"), []
            else:
                y, extra = ops.spade_norm_act(x, m, act="lrelu"), []
            y.backward(gon)
            outs += [y.detach(), x.grad] + ([g.grad, b.grad] if kind != "spade" else [m.grad]) + extra
        return outs
    v["norm"] = norm

    # warp forward / backward (gather through the inverted tap list), up-sampling, pooling, L1, SSIM
    img, flw, occ, gow = _rnd(120, 40, 64, 32, 64), _rnd(121, 40, 2, 32, 64, scale=3.0), torch.rand(40, 1, 32, 64).to(DEV), _rnd(122, 40, 64, 32, 64)

    def warp():
        i, f = img.clone().requires_grad_(True), flw.clone().requires_grad_(True)
        y = ops.flow_warp(i, f, occ)
        y.backward(gow)
        return [y.detach(), i.grad, f.grad]
    v["warp"] = warp

    xa, xb, gou = _rnd(130, 40, 64, 32, 64), _rnd(131, 40, 3, 128, 256).sigmoid(), _rnd(132, 40, 64, 64, 128)
    tgt = _rnd(133, 40, 3, 128, 256).sigmoid()

    def glue():
        a = xa.clone().requires_grad_(True)
        u = ops.upsample2x(a)
        p = ops.maxpool2x2(u)
        (u * gou).sum().backward(retain_graph=True)
        g1 = a.grad.clone()
        a.grad = None
        (p * xa).sum().backward()
        b = xb.clone().requires_grad_(True)
        l = ops.l1_mean(b, tgt) + ops.ssim_loss(b, tgt)
        l.backward()
        return [u.detach(), p.detach(), g1, a.grad, l.detach(), b.grad]
    v["glue"] = glue

    # fused multi-tensor Adam (c2m_adam_step)
    from c2m_amd.optim import Adam
    shapes = [(256, 256, 3, 3), (128, 64, 4, 4), (512,), (1024, 4096), (3, 32, 7, 7)]
    p0 = [_rnd(140 + i, *s) for i, s in enumerate(shapes)]
    g0 = [_rnd(150 + i, *s, scale=0.1) for i, s in enumerate(shapes)]

    def adam():
        ps = [torch.nn.Parameter(p.clone()) for p in p0]
        opt = Adam(ps, lr=2e-4, betas=(0.5, 0.999), eps=1e-7)
        for it in range(3):
            for p, g in zip(ps, g0):
                p.grad = g * (1.0 + it)
            opt.step()
        return [p.detach() for p in ps] + [opt.state[p]["exp_avg"] for p in ps] + [opt.state[p]["exp_avg_sq"] for p in ps]
    v["adam"] = adam
    return v


def make_aggressors(want):
    a = {}
    if "bf16_igemm" in want:
        with ops.conv_precision("bf16"):
            x = _rnd(200, 40, 128, 32, 64).bfloat16().requires_grad_(True)
            w = _rnd(201, 256, 128, 4, 4, scale=0.02).requires_grad_(True)

        def bf16_igemm(n):
            with ops.conv_precision("bf16"):
                for _ in range(n):
                    y = ops.conv(x, w, None, stride=2, padding=1, padding_mode="reflect", act="lrelu")
                    y.backward(y.detach())
                    x.grad = w.grad = None
        a["bf16_igemm"] = bf16_igemm
    if "bf16_igemm_fwd" in want:       # the bf16 gather kernel's forward launch alone (no data / weight gradient kernels)
        with ops.conv_precision("bf16"):
            xf = _rnd(202, 40, 128, 32, 64).bfloat16()
            wf = _rnd(203, 256, 128, 4, 4, scale=0.02)

        def bf16_igemm_fwd(n):
            with ops.conv_precision("bf16"), torch.no_grad():
                for _ in range(3 * n):
                    ops.conv(xf, wf, None, stride=2, padding=1, padding_mode="reflect", act="lrelu")
        a["bf16_igemm_fwd"] = bf16_igemm_fwd
    for name, dt in (("torch_bf16_gemm", torch.bfloat16), ("torch_fp32_gemm", torch.float32), ("torch_fp16_gemm", torch.float16)):
        if name in want:               # somebody else's MFMA kernels (hipBLASLt / rocBLAS through torch.matmul)
            ga, gb_ = _rnd(230, 4096, 4096).to(dt), _rnd(231, 4096, 4096).to(dt)

            def gemm(n, ga=ga, gb_=gb_):
                for _ in range(n):
                    torch.matmul(ga, gb_)
            a[name] = gemm
    if "bf16_patch" in want:
        xp = _rnd(210, 40, 128, 64, 128).bfloat16().requires_grad_(True)
        wp = _rnd(211, 128, 128, 3, 3, scale=0.03).requires_grad_(True)

        def bf16_patch(n):
            with ops.conv_precision("bf16"):
                for _ in range(max(1, n // 2)):
                    y = ops.conv(xp, wp, None, stride=1, padding=1, padding_mode="reflect", act="relu")
                    y.backward(y.detach())
                    xp.grad = wp.grad = None
        a["bf16_patch"] = bf16_patch
    if "fp32_wino" in want:
        xw = _rnd(220, 40, 128, 64, 128).requires_grad_(True)
        ww = _rnd(221, 128, 128, 3, 3, scale=0.03).requires_grad_(True)

        def fp32_wino(n):
            for _ in range(max(1, n // 2)):
                y = ops.conv(xw, ww, None, stride=1, padding=1, padding_mode="zeros", act="relu")
                y.backward(y.detach())
                xw.grad = ww.grad = None
        a["fp32_wino"] = fp32_wino
    if "bandwidth" in want:
        big = [torch.randn(64 << 20, device=DEV) for _ in range(3)]

        def bandwidth(n):
            for _ in range(n):
                torch.add(big[0], big[1], out=big[2])
        a["bandwidth"] = bandwidth
    if "rccl" in want:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
        bucket = torch.randn(16 << 20, device=DEV)

        def rccl(n):
            for _ in range(n):
                dist.all_reduce(bucket, op=dist.ReduceOp.AVG)
        a["rccl"] = rccl
    return a


def _clone(ts):
    return [t.detach().clone() for t in ts]


def run_pair(victim, aggressor, reps=3, burst=12):
    """Returns a list of (rep, order, output index, differing elements, max |diff|) for every output that differed."""
    A, B = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        ref = _clone(victim())            # solo (also settles plans / packs / allocator pools of this stream)
        ref2 = _clone(victim())
    torch.cuda.synchronize()
    bad = []
    for i, (p, q) in enumerate(zip(ref, ref2)):
        if not torch.equal(p, q):
            bad.append((-1, "solo-repeat", i, int((p != q).sum()), float((p.float() - q.float()).abs().max())))
    if aggressor is None:
        return bad
    with torch.cuda.stream(B):
        aggressor(2)                      # the neighbour's own warm-up
    torch.cuda.synchronize()
    for rep in range(reps):
        for order in ("neighbour-first", "victim-first"):
            if order == "neighbour-first":
                with torch.cuda.stream(B):
                    aggressor(burst)
            with torch.cuda.stream(A):
                out = victim()
            with torch.cuda.stream(B):
                aggressor(burst)          # arrives while the victim's kernels are running / queued
            with torch.cuda.stream(A):
                out2 = victim()
            torch.cuda.synchronize()
            for o in (out, out2):
                for i, (p, q) in enumerate(zip(ref, o)):
                    if not torch.equal(p, q):
                        bad.append((rep, order, i, int((p != q).sum()), float((p.float() - q.float()).abs().max())))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--victims", default="")
    ap.add_argument("--aggressors", default="bf16_igemm,bf16_patch,fp32_wino,bandwidth,rccl")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--burst", type=int, default=12)
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    vic = make_victims()
    if args.list:
        print("victims:", ", ".join(vic))
        return 0
    names = [n for n in args.victims.split(",") if n] or list(vic)
    agg = make_aggressors([n for n in args.aggressors.split(",") if n])
    from c2m_amd import _lib
    print("library:", _lib.LIB_PATH, flush=True)
    total = 0
    for vn in names:
        for an, af in agg.items():
            bad = run_pair(vic[vn], af, args.reps, args.burst)
            total += len(bad)
            print(f"{vn:14s} next to {an:11s}: {'bit-identical' if not bad else 'DIFFERS ' + str(bad[:6])}", flush=True)
    print("outputs that differed:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
