"""Where the GPU time OUTSIDE the conv MFMA kernels comes from: torch.profiler with Python stacks over a few steps of a bench
configuration, device time per (kernel, innermost c2m_amd source line).  For the fusions of DESIGN.md 5 (non-conv budget).
    python tools/gpu_time_by_source.py [--config 1] [--steps 3] [--top 70]
"""
import argparse, collections, copy, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from c2m_amd import ops
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_stream_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--top", type=int, default=70)
a = ap.parse_args()
c = bench.CONFIGS[a.config]
cfg = bench.bench_config(c["height"], c["width"], c["full_step"])
ops.set_conv_precision("bf16" if c["dtype"] == "bf16" else "fp32")
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to("cuda:0").train()
step = TrainStep(model, run_optimizers=c["full_step"], distributed=False)
batch = batch_to(make_stream_batch(c["batch"], c["windows"], c["height"], c["width"], 2, seed=0), "cuda:0")
batch["rng"] = {k: v.to("cuda:0") for k, v in make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0).items()}
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(a.steps):
        step(batch)
    torch.cuda.synchronize()

CONV = re.compile(r"reflect_ring_dgrad|conv_patch_nc8|conv_gather_nc8|conv_wgrad_nc8|conv_s2_dgrad_nc8|conv_wino_kernel|conv_wino4_kernel|conv_wino_wgrad_kernel|conv_igemm_kernel|conv_wgrad_kernel|conv_patch3x3|conv_wgrad_wide")
by = collections.defaultdict(lambda: [0.0, 0])
tot_conv = tot_other = 0.0
n_other = 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CUDA and not ev.kernels:
        continue
    for k in ev.kernels:
        dur = k.duration
        if CONV.search(k.name):
            tot_conv += dur
            continue
        src = "?"
        for fr in ev.stack:
            if "c2m_amd" in fr and "ops.py" not in fr and "_lib.py" not in fr:
                src = fr.split("c2m_amd/")[-1]
                break
        else:
            for fr in ev.stack:
                if "c2m_amd" in fr:
                    src = fr.split("c2m_amd/")[-1]
                    break
        if src == "?":                       # autograd thread: no Python frames; name the enclosing backward node
            q, names = ev, []
            while q is not None:
                names.append(q.name)
                q = q.cpu_parent
            src = "op: " + " < ".join(n.replace("autograd::engine::evaluate_function: ", "") for n in names[:3])
        kn = re.sub(r"<.*", "", k.name.replace("void ", ""))[:44]
        e = by[(kn, src[:80])]
        e[0] += dur; e[1] += 1
        tot_other += dur; n_other += 1
print(f"per step: conv MFMA kernels {tot_conv / a.steps / 1e3:.2f} ms, other kernels {tot_other / a.steps / 1e3:.2f} ms in "
      f"{n_other / a.steps:.0f} launches")
for (kn, src), (us, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"{us / a.steps / 1e3:7.3f} ms {n / a.steps:6.1f}x  {kn:44s} {src}")
