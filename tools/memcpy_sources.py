"""Who issues the hipMemcpy / hipMemset calls of a bench step (each is a ~4 us blit kernel on the GPU): torch.profiler runtime events
with Python stacks, counted per innermost c2m_amd source line or enclosing autograd op.
    python tools/memcpy_sources.py [--config 1] [--steps 2]"""
import argparse, collections, copy, os, sys
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
ap.add_argument("--steps", type=int, default=2)
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
by = collections.Counter()
for ev in prof.events():
    if "Memcpy" not in ev.name and "Memset" not in ev.name and "memcpy" not in ev.name and "memset" not in ev.name:
        continue
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    src = "?"
    q = ev
    while q is not None and src == "?":
        for fr in q.stack:
            if "c2m_amd" in fr and "_lib.py" not in fr:
                src = fr.split("c2m_amd/")[-1]
                break
        q = q.cpu_parent
    q, names = ev, []
    while q is not None:
        names.append(q.name.replace("autograd::engine::evaluate_function: ", ""))
        q = q.cpu_parent
    by[(ev.name, src[:70], " < ".join(names[1:4])[:90])] += 1
tot = sum(by.values())
print(f"{tot / a.steps:.0f} runtime memcpy / memset calls per step")
for (name, src, ops_), n in by.most_common(60):
    print(f"{n / a.steps:6.1f}x  {name:24s} {src:70s} {ops_}")
