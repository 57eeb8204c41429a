#!/usr/bin/env python
"""bench.py -- generator train-step frames/sec at 128x256x7 on 1/2/4/8 MI355X (BASELINE.json metric).

Default workload = BASELINE.json configs[1]: c2m_journal_cityscapes surface, 128x256, 7-frame clips (num_input_frames=2 +
5 predicted), per-GPU batch 8, fp32, generator forward + backward (both discriminators off, VGG perceptual loss on),
synthetic data + random-init weights, weak scaling over ranks with the mean-of-ranks gradient all-reduce (RCCL).
A step = zero_grad + forward + backward (+ gradient all-reduce when N > 1); inputs are resident in HBM.
`--config 2|3|4` select the other BASELINE configurations (see CONFIGS below).

Launch: `python bench.py --gpus N` starts N ranks ITSELF (one process per GPU through torch.distributed.run, RCCL over
xGMI) when it is not already running under a launcher; under `python -m torch.distributed.run ... bench.py --gpus N` it
joins the existing rendezvous.  A world size different from --gpus is an error, never a silent 1-rank run.

One JSON line on rank 0.  Extra objects:
  roofline     all conv MFMA launches (Winograd, implicit-GEMM gather / LDS-patch, weight gradient), HIP-event timed on
               the launch stream inside the timed steps.  `achieved` = EXECUTED MFMA FLOP/s (algorithmic direct-conv
               FLOPs on the unpadded domain; Winograd launches divided by their 2.25x multiply reduction) against the
               fp32 matrix peak; per-family and whole-step figures beside it.
  cpu_baseline the CPU oracle (oracle/c2m_oracle.py, validated bit-exact against the reference) timed on this host
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from c2m_amd import ops  # noqa: E402
from c2m_amd.config import default_config, normalize_config  # noqa: E402
from c2m_amd.modules.model import GeneratorFullModel  # noqa: E402
from c2m_amd.synthetic import make_stream_batch, make_batch, make_step_rng, batch_to  # noqa: E402
from c2m_amd.train import TrainStep, init_distributed  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD, 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16
WINOGRAD_REDUCTION = 2.25         # F(2x2,3x3): 16 multiplies per 4 outputs instead of 36
WINOGRAD4_REDUCTION = 4.0         # F(4x4,3x3): 36 multiplies per 16 outputs instead of 144
# SURVEY.md 8d, FlopCounterMode on the reference graph, per 7-frame clip at 128x256 (conv FLOPs scale with the pixels)
ALGO_GFLOP_PER_CLIP = {False: 1068.6, True: 1159.8}          # [full_step]

# BASELINE.json configs[k] -> flags (configs[0] is the CPU reference case = the cpu_baseline leg of this file)
CONFIGS = {
    1: dict(height=128, width=256, batch=8, windows=1, dtype="f32", full_step=False),
    2: dict(height=256, width=512, batch=4, windows=1, dtype="bf16", full_step=True),
    3: dict(height=128, width=256, batch=8, windows=1, dtype="bf16", full_step=True),     # global batch 64 at 8 ranks
    4: dict(height=256, width=512, batch=4, windows=2, dtype="bf16", full_step=True),     # 14-frame streams = 2 windows
}


def bench_config(height, width, full_step=False):
    return normalize_config(default_config(height=height, width=width, num_input_frames=2,
                                           use_image_discriminator=full_step, use_video_discriminator=full_step))


def _physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        n = len(seen)
    except OSError:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n or avail, avail))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, seconds_budget=30.0):
    """Oracle fwd + bwd of ONE clip (B=1) on the host cores; the same graph the GPU runs, generator-only.
    A B=1 graph does not scale to a 128-core host (more threads = slower), so a short sweep picks the thread count with
    the best throughput, then >= 3 steps are timed at that count."""
    from oracle import c2m_oracle as O
    import copy
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"],
                               model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    del model
    h, w = cfg["train_params"]["input_size"]
    batch = make_batch(1, h, w, 2, seed=0)
    rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)

    def one_step():
        S = O.State(sd)
        b = dict(batch)
        b["tracking_gnn"] = batch["tracking_gnn"].clone()
        t0 = time.time()
        _, lg, _, _ = O.forward(S, cfg, b, rng)
        O.train_step_backward(cfg, lg, {}, {})
        return time.time() - t0

    phys = _physical_cores()
    prev = torch.get_num_threads()
    t_start = time.time()
    sweep = {}
    for nt in sorted({n for n in (8, 16, 32, 64, phys) if n <= phys}):
        torch.set_num_threads(nt)
        if not sweep:
            one_step()                           # warm-up (allocator, oneDNN primitive caches)
        sweep[nt] = one_step()
        if time.time() - t_start > 0.5 * seconds_budget:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]]
    while len(times) < 3 or (len(times) < 6 and time.time() - t_start < seconds_budget):
        times.append(one_step())
    torch.set_num_threads(prev)
    times.sort()
    med = times[len(times) // 2]
    # host-saturating figure: P independent replicas (one clip each) x the best thread count, all running at once
    saturated = None
    # at most 4 replicas: the GPU boxes' process guard allows 6 processes with the device open, and a child that imports
    # torch counts as one (this parent is the fifth) -- so on a 128-core host this leg loads 4 x `best` cores, not all
    P = min(phys // best, 4)
    if P > 1:
        h_, w_ = cfg["train_params"]["input_size"]
        env = dict(os.environ, OMP_NUM_THREADS=str(best), MKL_NUM_THREADS=str(best), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(best), "--height", str(h_), "--width", str(w_)]
        procs = [subprocess.Popen(cmd + ["--cpu-worker-index", str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True, env=env) for r in range(P)]       # replica r is pinned to its own `best` cores
        per = []
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=240)
                per += [float(l.split()[1]) for l in out.splitlines() if l.startswith("CPU_WORKER_S_PER_STEP")]
            except Exception:
                pr.kill()
        if len(per) == P:
            per.sort()
            saturated = {"value": round(P * 7.0 / per[len(per) // 2], 3), "unit": "frames/s", "processes": P,
                         "threads_each": best, "cores": P * best,
                         "sample": f"{P} concurrent replicas x 2 timed steps of 1 clip each, median replica {per[len(per) // 2]:.3f} s/step"}
    return {"value": round(7.0 / med, 3), "unit": "frames/s", "cores": best, "kind": "port", "multi_process": saturated,
            "host": f"{_cpu_model()}, {phys} physical cores visible",
            "thread_sweep_s_per_step": {str(k): round(v, 3) for k, v in sweep.items()},
            "reference_code_datapoint": "the reference itself, build container (8 vCPU): 3.0 frames/s G-only (SURVEY.md 6)",
            "sample": f"{len(times)} timed steps of 1 clip (B=1, 7 frames, {h}x{w}, G-only fwd+bwd) at {best} threads "
                      f"(best of the sweep); median {med:.3f} s/step"}


def _cpu_worker(nthreads, height, width, index=0):
    """One replica of the multi-process cpu_baseline leg: oracle fwd + bwd of one clip, `nthreads` threads pinned to the
    replica's own slice of the allowed CPUs (unpinned replicas migrate onto each other's cores); prints s/step."""
    from oracle import c2m_oracle as O
    import copy
    if hasattr(os, "sched_setaffinity"):
        cpus = sorted(os.sched_getaffinity(0))
        mine = cpus[index * nthreads:(index + 1) * nthreads]
        if len(mine) == nthreads:
            os.sched_setaffinity(0, set(mine))
    torch.set_num_threads(nthreads)
    cfg = bench_config(height, width, False)
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"],
                               model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    del model
    batch = make_batch(1, height, width, 2, seed=0)
    rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
    ts = []
    for it in range(3):
        S = O.State(sd)
        b = dict(batch)
        b["tracking_gnn"] = batch["tracking_gnn"].clone()
        t0 = time.time()
        _, lg, _, _ = O.forward(S, cfg, b, rng)
        O.train_step_backward(cfg, lg, {}, {})
        if it:
            ts.append(time.time() - t0)
    print(f"CPU_WORKER_S_PER_STEP {sum(ts) / len(ts):.4f}", flush=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(n):
    """Not under a launcher: start n ranks (one per GPU) as children and return their exit code.  This process has not
    touched the GPU (no HIP call before this point), and it never execs: the children are fresh processes."""
    have = torch.cuda.device_count()              # does not initialise the GPU
    if have < n and not os.environ.get("C2M_REHEARSAL_SHARED_GPU"):   # rehearsal: all ranks share GPU 0 over gloo (tests)
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _family(s, names, divide=1.0, peak=PEAK_FP32_MFMA_TFLOPS, step_s=None):
    fl = sum(s[n]["flops"] for n in names if n in s)
    ms = sum(s[n]["ms"] for n in names if n in s)
    nl = sum(s[n]["launches"] for n in names if n in s)
    if nl == 0:
        return None
    alg = fl / (ms * 1e-3) / 1e12
    return {"launches": nl, "ms": round(ms, 3), "algorithmic_tflops": round(alg, 2),
            "executed_mfma_tflops": round(alg / divide, 2), "frac": round(alg / divide / peak, 4),
            "share_of_step_time": round(ms * 1e-3 / step_s, 3) if step_s else None}


def side_config(k, dev, steps=10, warmup=3):
    """One of the bf16 BASELINE configurations (2: 256x512 B4, 3: 128x256 B8; full adversarial step) measured in THIS process after
    (and outside) the headline timed region: `warmup` eager steps, one eager step with HIP events around every conv launch (the conv
    fraction of the bf16 matrix peak), then `steps` HIP-graph replays of zero_grad + forward + backward with the four Adam steps
    eager behind each replay, bracketed by synchronize().  Same code path as `bench.py --config k` (its `hip_graph_replay` object)."""
    import copy
    import gc
    c = CONFIGS[k]
    cfg = bench_config(c["height"], c["width"], c["full_step"])
    prev = ops.set_conv_precision("bf16" if c["dtype"] == "bf16" else "fp32")
    try:
        torch.manual_seed(0)
        model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"],
                                   model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
        model.to(dev).train()
        step = TrainStep(model, run_optimizers=c["full_step"], distributed=False)
        clips = c["batch"] * c["windows"]
        batch = batch_to(make_stream_batch(c["batch"], c["windows"], c["height"], c["width"], 2, seed=0), dev)
        rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
        batch["rng"] = {kk: v.to(dev) for kk, v in rng.items()}
        with torch.cuda.stream(step.graph_stream):
            for _ in range(warmup):
                step(batch)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            step(batch)
            torch.cuda.synchronize(dev)
            eager_ms = 1000.0 * (time.perf_counter() - t0)
            prof = ops.ConvProfiler()
            ops.ConvProfiler.active = prof
            step(batch)
            ops.ConvProfiler.active = None
            s = prof.summary()
            step.capture(batch)
            step(batch)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step(batch)
            torch.cuda.synchronize(dev)
            tg = time.perf_counter() - t0
        bf = c["dtype"] == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
        fl = sum(d["flops"] for d in s.values())
        ms = sum(d["ms"] for d in s.values())
        fwd = _family(s, ["igemm_bf16" if bf else "igemm"], 1.0, peak)
        wg = _family(s, ["wgrad_bf16" if bf else "wgrad"], 1.0, peak)
        out = {"workload": f"BASELINE configs[{k}]: {c['height']}x{c['width']}, 7-frame clips, batch {c['batch']}/GPU"
                           f"{'' if c['windows'] == 1 else ' x %d windows' % c['windows']}, "
                           f"{'bf16 data path' if bf else 'fp32'}, full adversarial step (G + D_image + D_video, 4 Adam steps), VGG loss on",
               "ms_per_step": round(1000.0 * tg / steps, 3), "value": round(clips * 7 * steps / tg, 2), "unit": "frames/s",
               "steps": steps, "warmup": warmup, "mode": "HIP-graph replay of zero_grad + forward + backward, optimizers eager",
               "eager_ms_per_step": round(eager_ms, 3),
               "conv_launches_per_step": sum(d["launches"] for d in s.values()), "conv_ms_per_step": round(ms, 3),
               "conv_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "conv_frac_of_peak": round(fl / (ms * 1e-3) / 1e12 / peak, 4),
               "peak_tflops": peak,
               "conv_fwd_dgrad_tflops": fwd and fwd["algorithmic_tflops"], "conv_wgrad_tflops": wg and wg["algorithmic_tflops"],
               "conv_timing": "HIP events around every conv launch of ONE eager step (layout passes of a launch's input included)"}
        del step, model, batch, prof
        return out
    finally:
        ops.set_conv_precision(prev)
        gc.collect()
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1,
                    help="BASELINE.json configs[k]: 1 = the metric's configuration (default); 2 = 256x512 B4 bf16 full step; "
                         "3 = 128x256 B8/rank bf16 full step; 4 = 256x512, 2 x 7-frame windows, bf16 full step")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU")
    ap.add_argument("--windows", type=int, default=None,
                    help="7-frame windows per stream sample (2 = BASELINE configs[4]'s 14-frame streams); clips/GPU = batch x windows")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-table", default=None, help="write a per-shape conv timing table to this file")
    ap.add_argument("--event-every", type=int, default=10,
                    help="the HIP events around every conv launch (roofline object) are recorded in every k-th step of the timed "
                         "region.  Those steps time ISOLATED launches: the branch streams of ops.aux_branch / ops.deferred_wgrads are "
                         "off in them (one stream, ~3 ms more than a normal step) and ~600 event records cost ~1 ms; 1 = every step")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default=None,
                    help="conv operand precision: f32 = BASELINE configs[1] (the bench line); bf16 = configs[2-4] mode")
    ap.add_argument("--full-step", action="store_true", default=None,
                    help="full adversarial step of BASELINE configs[2-4]: both discriminators on + the four Adam steps")
    ap.add_argument("--check-grads", action="store_true",
                    help="after the run, verify that every rank holds bit-identical (all-reduced) gradients "
                         "(always on when N > 1)")
    ap.add_argument("--graph", action="store_true",
                    help="side measurement: capture zero_grad + forward + backward into a HIP graph and time replays (no "
                         "per-kernel events, so no roofline object; with N > 1 the all-reduces run eagerly after each replay)")
    ap.add_argument("--graph-side", dest="graph_side", action="store_true", default=None,
                    help="also time HIP-graph replays of the step after the eager timed region (default: on for configs 2-4)")
    ap.add_argument("--no-graph-side", dest="graph_side", action="store_false")
    ap.add_argument("--fp32-allreduce", action="store_true",
                    help="keep fp32 gradient buckets on the wire in the bf16 configurations (default there: bf16 copies)")
    ap.add_argument("--measure-comm", action="store_true",
                    help="also record one HIP event per gradient bucket and step (allreduce_exposed_ms_per_bucket); off by default: "
                         "the timed region then carries only the two events of allreduce_exposed_ms_per_step")
    ap.add_argument("--force-reducer", action="store_true",
                    help="run the bucketed RCCL all-reduce path even with one rank (plumbing check on a single GPU)")
    ap.add_argument("--side-configs", default=None,
                    help="comma-separated BASELINE configs measured as HIP-graph replays in the same process AFTER the headline timed "
                         "region and reported as `side_configs` (default: '3,2' for the plain 1-GPU configs[1] run, '' otherwise)")
    ap.add_argument("--side-steps", type=int, default=10)
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-index", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return _cpu_worker(args.cpu_worker, args.height or 128, args.width or 256, args.cpu_worker_index)
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))
    if args.force_reducer:
        os.environ["C2M_FORCE_PROCESS_GROUP"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    rank, local_rank, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or run plain `python bench.py --gpus {args.gpus}`)")
    if world > 1 and not dist.is_initialized():
        raise SystemExit("bench.py: WORLD_SIZE > 1 but no process group (MASTER_ADDR / MASTER_PORT missing)")
    dev = torch.device("cuda", local_rank)
    cfg = bench_config(args.height, args.width, args.full_step)
    ops.set_conv_precision("bf16" if args.dtype == "bf16" else "fp32")
    import copy
    torch.manual_seed(0)          # same seed on every rank; the reducer ALSO broadcasts rank 0's parameters and buffers
    model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"],
                               model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
    model.to(dev).train()
    distributed = world > 1 or args.force_reducer
    # bf16 configurations all-reduce bf16 copies of the fp32 gradient buckets (212 MB instead of 424 MB)
    comm_dtype = torch.bfloat16 if (args.dtype == "bf16" and not args.fp32_allreduce) else torch.float32
    step = TrainStep(model, run_optimizers=args.full_step, distributed=distributed,
                     force_collectives=args.force_reducer, measure_comm=True, comm_dtype=comm_dtype,
                     measure_comm_buckets=args.measure_comm)
    clips = args.batch * args.windows
    batch = batch_to(make_stream_batch(args.batch, args.windows, args.height, args.width, 2, seed=rank), dev)
    rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=rank)
    batch["rng"] = {k: v.to(dev) for k, v in rng.items()}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    import contextlib
    # configs[2-4] (full adversarial step, ~2000 launches): the eager line is partly host-bound, so the same run also times
    # HIP-graph replays of zero_grad + forward + backward (+ eager optimizers / all-reduce) as a side object
    graph_side = args.graph_side if args.graph_side is not None else (args.config != 1 and not args.graph)
    gctx = torch.cuda.stream(step.graph_stream) if (args.graph or graph_side) else contextlib.nullcontext()
    graph_side_result = None
    with gctx:                                  # with --graph every step (eager warm-up included) runs on the capture stream
        for _ in range(args.warmup):
            step(batch)
        if args.graph:
            args.no_roofline = True
            step.capture(batch)
            step(batch)
        if step.reducer is not None:
            step.reducer.reset_measurements()
        barrier()
        prof = None if args.no_roofline else ops.ConvProfiler()
        t0 = time.perf_counter()
        sampled = 0
        for i in range(args.steps):
            on = prof is not None and i % max(args.event_every, 1) == 0
            ops.ConvProfiler.active = prof if on else None
            sampled += int(on)
            step(batch)
        ops.ConvProfiler.active = None
        barrier()
        elapsed = time.perf_counter() - t0
        if graph_side:
            step.capture(batch)
            step(batch)
            barrier()
            tg = time.perf_counter()
            for _ in range(args.steps):
                step(batch)
            barrier()
            tg = time.perf_counter() - tg
            if world > 1:
                t = torch.tensor([tg], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tg = float(t.item())
            graph_side_result = {"ms_per_step": round(1000.0 * tg / args.steps, 3),
                                 "value": round(world * clips * 7 * args.steps / tg, 2), "unit": "frames/s",
                                 "what": "the same step as a HIP-graph replay (zero_grad + forward + backward captured; optimizers"
                                         " and the gradient all-reduce run eagerly after each replay), same process, same batch"}
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = world * clips * 7 * args.steps
    step_s = elapsed / args.steps
    tflop_per_step = ALGO_GFLOP_PER_CLIP[bool(args.full_step)] * (args.height * args.width) / (128 * 256) * 1e-3 * clips
    prec = "fp32" if args.dtype == "f32" else "bf16 data path (bf16 activations and conv operands, fp32 accumulate / weights / losses)"
    result = {
        "metric": f"generator train-step frames/sec at {args.height}x{args.width}x7", "value": round(frames / elapsed, 2),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * step_s, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{args.config}]: {args.height}x{args.width}, 7-frame clips (2 in + 5 predicted), "
                               f"batch {args.batch}/GPU{'' if args.windows == 1 else f' x {args.windows} windows (14-frame streams)'}, {prec}, "
                               + ("full adversarial step (G + D_image + D_video, 4 Adam steps), VGG loss on, "
                                  if args.full_step else "generator fwd+bwd only (no D), VGG loss on, ")
                               + "random-init weights", "global_batch": world * clips,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        "rccl_ranks": world if dist.is_initialized() and dist.get_backend() == "nccl" else 0,
        "hip_graph": bool(args.graph),
        "hip_graph_replay": graph_side_result,
        "achieved_tflops_algorithmic": round(tflop_per_step * world / step_s, 2),
    }
    if step.reducer is not None:
        result["allreduce_bytes_per_step"] = step.reducer.bytes_per_step()
        result["allreduce_buckets"] = len(step.reducer.buckets)
        result["allreduce_dtype"] = "bf16" if comm_dtype == torch.bfloat16 else "f32"
        result["allreduce_op"] = "avg" if step.reducer.use_avg else "pre-divide + sum"
        ex = step.reducer.exposed_ms()
        if ex is not None and world > 1:
            t = torch.tensor([ex], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ex = float(t.item())
        result["allreduce_exposed_ms_per_step"] = None if ex is None else round(ex, 3)
        pb = step.reducer.exposed_ms_per_bucket()     # rank 0's view: (bucket, ms its reduced gradients arrived after backward / the previous bucket)
        result["allreduce_exposed_ms_per_bucket"] = None if pb is None else [[int(b), round(v, 3)] for b, v in pb]
    if args.check_grads or world > 1:
        # all-reduced gradients must be bit-identical on every rank although each rank saw different data
        gsum = torch.stack([p.grad.double().abs().sum() for p in model.parameters() if p.grad is not None])
        local = torch.stack([gsum.sum(), (gsum * torch.arange(1, gsum.numel() + 1, device=dev)).sum()])
        if dist.is_initialized() and world > 1:
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            same = all(torch.equal(g, gathered[0]) for g in gathered)
        else:
            same = True
        if not same:
            raise SystemExit(f"rank {rank}: gradients differ across ranks after the all-reduce")
        result["grad_sync"] = f"bit-identical gradient fingerprints on all {world} rank(s)" if distributed else "single rank, no reducer"
    if rank == 0:
        if prof is not None:
            s = prof.summary()
            ev_s = elapsed * sampled / args.steps          # wall time of the steps that carried events
            bf = args.dtype == "bf16"
            sfx = "_bf16" if bf else ""
            peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
            fam = {
                "winograd_fwd_dgrad (conv_wino_kernel)": _family(s, ["wino"], WINOGRAD_REDUCTION, PEAK_FP32_MFMA_TFLOPS, ev_s),
                "winograd_f4x4_fwd_dgrad (conv_wino4_kernel)": _family(s, ["wino4"], WINOGRAD4_REDUCTION, PEAK_FP32_MFMA_TFLOPS, ev_s),
                "direct_fwd_dgrad (conv_igemm_kernel, conv_patch3x3_kernel)": _family(s, ["igemm" + sfx], 1.0, peak, ev_s),
                "wgrad (conv_wgrad_kernel)": _family(s, ["wgrad" + sfx], 1.0, peak, ev_s),
                "winograd_wgrad (conv_wino_wgrad_kernel)": _family(s, ["wino_wgrad"], WINOGRAD_REDUCTION, PEAK_FP32_MFMA_TFLOPS, ev_s),
            }
            fam = {k: v for k, v in fam.items() if v}
            ms_all = sum(v["ms"] for v in fam.values())
            exe = sum(v["executed_mfma_tflops"] * v["ms"] for v in fam.values()) / ms_all
            alg = sum(v["algorithmic_tflops"] * v["ms"] for v in fam.values()) / ms_all
            nl = sum(v["launches"] for v in fam.values())
            nbytes = sum(d["bytes"] for d in s.values())
            # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command
            # (tools/pmc_traffic.sh -> profiles/r02_pmc_traffic.json); counters cannot be read from inside the process
            # of the SAME build: the file records the conv launches per step it was taken at, and a different count here
            # (a kernel was added / fused since) makes it stale -> traffic null, with the reason
            traffic, traffic_src = None, None
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            if cands and args.config == 1 and (args.batch, args.windows, args.height, args.width) == (8, 1, 128, 256) \
                    and args.dtype == "f32" and not args.full_step:
                with open(cands[-1]) as f:
                    tj = json.load(f)
                taken_at = tj["conv"].get("launches_per_step")
                if taken_at is not None and abs(float(taken_at) - nl / sampled) < 0.5:
                    traffic, traffic_src = tj["conv"]["traffic_bytes_per_launch"], os.path.basename(cands[-1]) + ": " + tj["note"]
                else:
                    traffic_src = (f"{os.path.basename(cands[-1])} is stale: taken at {taken_at} conv launches/step, this build "
                                   f"runs {nl / sampled:.1f}; regenerate with tools/prof_round.sh")
            result["roofline"] = {
                "bound": "mfma",
                "kernel": "all conv MFMA launches: conv_wino_kernel + conv_wino4_kernel + conv_igemm_kernel + conv_patch3x3_kernel "
                          "(forward, data gradient) + conv_wgrad_kernel + conv_wino_wgrad_kernel (weight gradient)",
                "achieved": round(exe, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(exe / peak, 4),
                "flops_counted": "EXECUTED MFMA FLOPs = algorithmic direct-convolution FLOPs (2*M*K*Npix on the unpadded "
                                 "domain) with F(2x2,3x3) Winograd launches divided by 2.25 and F(4x4,3x3) launches by 4",
                "algorithmic_tflops": round(alg, 2),
                "families": fam,
                "whole_step": {"algorithmic_tflop_per_step": round(tflop_per_step, 3),
                               "achieved": round(tflop_per_step / step_s, 2), "frac": round(tflop_per_step / step_s / peak, 4),
                               "definition": "SURVEY 8d: FlopCounterMode FLOPs of the reference graph / step wall time"},
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(nbytes / max(nl, 1)),
                "launches_per_step": nl // sampled,
                "event_steps": f"{sampled} of the {args.steps} timed steps carried the HIP events (every {max(args.event_every, 1)}-th)",
                "avg_launch_us": round(1000.0 * ms_all / max(nl, 1), 2),
                "share_of_step_time": round(ms_all / (1000.0 * ev_s), 3),
                # steps are issued asynchronously, so an event step has no wall time of its own: the shares divide by the MEAN
                # step time x event steps; an event step is ~1 ms (1.5 %) longer than a plain one, shares are biased up by that
                "share_basis": "mean wall time per step x event steps (event steps run on one stream and carry the events: ~4 ms longer than the mean step; shares biased up by <= 7 %)",
            }
        if prof is not None and args.conv_table:
            with open(args.conv_table, "w") as f:
                f.write("kind pass M K Npix taps stride reflect | launches/step ms/step TFLOP/s\n")
                for tag, n, ms, tf in prof.table():
                    f.write(f"{tag} | {n / sampled:.1f} {ms / sampled:.3f} {tf:.1f}\n")
        side = args.side_configs
        if side is None:
            side = "3,2" if (world == 1 and args.config == 1 and not args.graph and not args.force_reducer and
                             (args.batch, args.windows, args.height, args.width, args.dtype) == (8, 1, 128, 256, "f32")) else ""
        if side and world == 1:
            # the bf16 configurations, outside the timed region (after it): the headline objects must not hold HBM while they run
            import gc
            del step, model, batch
            prof = None
            gc.collect()
            torch.cuda.empty_cache()
            result["side_configs"] = {}
            for k in [int(v) for v in side.split(",") if v.strip()]:
                try:
                    result["side_configs"][f"configs[{k}]"] = side_config(k, dev, steps=args.side_steps)
                except Exception as e:                     # a side measurement never takes the headline line down
                    result["side_configs"][f"configs[{k}]"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(bench_config(args.height, args.width, False))
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
