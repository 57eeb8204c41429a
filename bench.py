#!/usr/bin/env python
"""bench.py -- generator train-step frames/sec at 128x256x7 on 1/2/4/8 MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): c2m_journal_cityscapes surface, 128x256, 7-frame clips (num_input_frames=2 +
5 predicted), per-GPU batch 8, fp32, generator forward + backward (both discriminators off, VGG perceptual loss on),
synthetic data + random-init weights, weak scaling over ranks with the mean-of-ranks gradient all-reduce (RCCL).
A step = zero_grad + forward + backward (+ gradient all-reduce when N > 1); inputs are resident in HBM.

One JSON line on rank 0.  Extra objects:
  roofline     dominant kernels = conv_wino_kernel + conv_igemm_kernel + conv_patch3x3_kernel (fp32 MFMA; forward and
               data-gradient launches): algorithmic (direct-convolution) FLOPs / HIP-event time of those launches,
               measured inside the timed steps on the launch stream; peak = 157.3 TFLOP/s fp32 matrix.  The Winograd
               kernel executes 2.25x fewer MFMA FLOPs than the algorithmic count on the layers it covers.
  cpu_baseline the CPU oracle (oracle/c2m_oracle.py, validated bit-exact against the reference) timed on this host
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from c2m_amd import ops  # noqa: E402
from c2m_amd.config import default_config, normalize_config  # noqa: E402
from c2m_amd.modules.model import GeneratorFullModel  # noqa: E402
from c2m_amd.synthetic import make_batch, make_stream_batch, make_step_rng, batch_to  # noqa: E402
from c2m_amd.train import TrainStep, init_distributed  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 (only used by the --dtype bf16 side measurement)
ALGO_GFLOP_PER_CLIP = 1068.6      # 7-frame 128x256 G-only clip, fwd+bwd (FlopCounterMode on the reference)


def bench_config(height, width, full_step=False):
    return normalize_config(default_config(height=height, width=width, num_input_frames=2,
                                           use_image_discriminator=full_step, use_video_discriminator=full_step))


def cpu_baseline(cfg, seconds_budget=25.0):
    """Oracle fwd + bwd of ONE clip (B=1) on the host cores; the same graph the GPU runs, generator-only."""
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
    times = []
    t_start = time.time()
    for it in range(6):
        S = O.State(sd)
        b = dict(batch)
        b["tracking_gnn"] = batch["tracking_gnn"].clone()
        t0 = time.time()
        _, lg, _, _ = O.forward(S, cfg, b, rng)
        O.train_step_backward(cfg, lg, {}, {})
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and len(times) >= 2:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(7.0 / med, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} timed steps (after 1 warm-up) of 1 clip (B=1, 7 frames, {h}x{w}, G-only fwd+bwd); "
                      f"median {med:.3f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--windows", type=int, default=1,
                    help="7-frame windows per stream sample (2 = BASELINE configs[4]'s 14-frame streams); clips/GPU = batch x windows")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-table", default=None, help="write a per-shape conv timing table to this file")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="conv operand precision: f32 = BASELINE configs[1] (the bench line); bf16 = configs[2-4] mode")
    ap.add_argument("--full-step", action="store_true",
                    help="full adversarial step of BASELINE configs[2-4]: both discriminators on + the four Adam steps")
    ap.add_argument("--check-grads", action="store_true",
                    help="after the run, verify that every rank holds bit-identical (all-reduced) gradients")
    ap.add_argument("--force-reducer", action="store_true",
                    help="run the bucketed RCCL all-reduce path even with one rank (plumbing check on a single GPU)")
    args = ap.parse_args()

    if "--force-reducer" in sys.argv:
        os.environ["C2M_FORCE_PROCESS_GROUP"] = "1"
    rank, local_rank, world = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank)
    cfg = bench_config(args.height, args.width, args.full_step)
    ops.set_conv_precision("bf16" if args.dtype == "bf16" else "fp32")
    import copy
    torch.manual_seed(0)                       # identical initial weights on every rank (== DDP's rank-0 broadcast)
    model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"],
                               model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
    model.to(dev).train()
    step = TrainStep(model, run_optimizers=args.full_step, distributed=world > 1 or args.force_reducer,
                     force_collectives=args.force_reducer)
    clips = args.batch * args.windows
    batch = batch_to(make_stream_batch(args.batch, args.windows, args.height, args.width, 2, seed=rank), dev)
    rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=rank)
    batch["rng"] = {k: v.to(dev) for k, v in rng.items()}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(batch)
    barrier()
    prof = None if args.no_roofline else ops.ConvProfiler()
    t0 = time.perf_counter()
    if prof is not None:
        prof.__enter__()
    for _ in range(args.steps):
        step(batch)
    if prof is not None:
        prof.__exit__(None, None, None)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = world * clips * 7 * args.steps
    result = {
        "metric": f"generator train-step frames/sec at {args.height}x{args.width}x7", "value": round(frames / elapsed, 2),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE {'configs[1]' if (args.dtype, args.full_step) == ('f32', False) else 'configs[2-3] style (side measurement)'}: "
                               f"{args.height}x{args.width}, 7-frame clips (2 in + 5 predicted), "
                               f"batch {args.batch}/GPU{'' if args.windows == 1 else f' x {args.windows} windows (14-frame streams)'}, {'fp32' if args.dtype == 'f32' else 'bf16 conv operands (fp32 accumulate, fp32 tensors)'}, "
                               + ("full adversarial step (G + D_image + D_video, 4 Adam steps), VGG loss on, "
                                  if args.full_step else "generator fwd+bwd only (no D), VGG loss on, ")
                               + "random-init weights", "global_batch": world * clips,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        # conv FLOPs scale with the pixel count ("for 256x512 multiply conv FLOPs by 4")
        "achieved_tflops_algorithmic": round(ALGO_GFLOP_PER_CLIP * (args.height * args.width) / (128 * 256) * 1e-3 *
                                             world * clips * args.steps / elapsed, 2),
    }
    if args.check_grads:
        # all-reduced gradients must be bit-identical on every rank although each rank saw different data
        gsum = torch.stack([p.grad.double().abs().sum() for p in model.parameters() if p.grad is not None])
        local = torch.stack([gsum.sum(), (gsum * torch.arange(1, gsum.numel() + 1, device=dev)).sum()]).cpu()
        if dist.is_initialized() and world > 1:
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            same = all(torch.equal(g, gathered[0]) for g in gathered)
        else:
            same = True
        if not same:
            raise SystemExit(f"rank {rank}: gradients differ across ranks after the all-reduce")
        result["grad_sync"] = "identical on all ranks"
    if rank == 0:
        if prof is not None:
            s = prof.summary()
            sfx = "_bf16" if args.dtype == "bf16" else ""
            peak = PEAK_BF16_MFMA_TFLOPS if sfx else PEAK_FP32_MFMA_TFLOPS
            ig = s.get("igemm" + sfx, dict(launches=0, flops=0.0, ms=1e-9, bytes=0.0))
            wg = s.get("wgrad" + sfx, dict(launches=0, flops=0.0, ms=1e-9, bytes=0.0))
            # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command
            # (tools/pmc_traffic.sh -> profiles/r01_pmc_traffic.json); counters cannot be read from inside the process
            traffic, traffic_src = None, None
            tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(tp) and args.windows == 1 and (args.batch, args.height, args.width, args.dtype, args.full_step) == \
                    (8, 128, 256, "f32", False):
                with open(tp) as f:
                    traffic = json.load(f)["igemm"]["traffic_bytes_per_launch"]
                traffic_src = "profiles/r01_pmc_traffic.json (FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
            ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
            result["roofline"] = {
                "bound": "mfma", "kernel": "conv_wino_kernel + conv_igemm_kernel + conv_patch3x3_kernel (conv forward + data-gradient launches)",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "flops_counted": "algorithmic direct-convolution FLOPs (2*M*K*Npix)",
                "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(ig["bytes"] / max(ig["launches"], 1)),
                "launches_per_step": ig["launches"] // args.steps,
                "avg_launch_us": round(1000.0 * ig["ms"] / max(ig["launches"], 1), 2),
                "gflop_per_launch": round(ig["flops"] / max(ig["launches"], 1) / 1e9, 3),
                "share_of_step_time": round(ig["ms"] / (1000.0 * elapsed), 3),
                "wgrad": {"achieved": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2),
                          "launches_per_step": wg["launches"] // args.steps,
                          "share_of_step_time": round(wg["ms"] / (1000.0 * elapsed), 3)},
            }
        if prof is not None and args.conv_table:
            with open(args.conv_table, "w") as f:
                f.write("kind pass M K Npix taps stride reflect | launches/step ms/step TFLOP/s\n")
                for tag, n, ms, tf in prof.table():
                    f.write(f"{tag} | {n / args.steps:.1f} {ms / args.steps:.3f} {tf:.1f}\n")
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
