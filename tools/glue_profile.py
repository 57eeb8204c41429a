"""Which torch ops (not our kernels) run in one bench step: counts and device time by op name + input shapes.
usage: python tools/glue_profile.py"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

dev = torch.device("cuda", 0)
cfg = bench.bench_config(128, 256, False)
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to(dev).train()
step = TrainStep(model, run_optimizers=False, distributed=False)
batch = batch_to(make_batch(8, 128, 256, 2, seed=0), dev)
rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
batch["rng"] = {k: v.to(dev) for k, v in rng.items()}
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
dt = lambda e: getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
rows.sort(key=lambda e: -dt(e))
print("self device us | calls | op | input shapes")
for e in rows[:40]:
    print(f"{dt(e):9.0f} {e.count:5d}  {e.key:32s} {str(e.input_shapes)[:110]}")
print("total aten self device ms:", sum(dt(e) for e in rows) / 1e3)
