"""Which ops produce the tensors that c2m_nchw_to_nc8 converts in one step of a bf16 bench configuration: bytes per producing
autograd node (forward) and in total for backward -- the candidates for writing NC8 directly in the producer.
    python tools/nc8_producers.py [--config 3]"""
import argparse, collections, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from c2m_amd import ops
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_stream_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
a = ap.parse_args()
c = bench.CONFIGS[a.config]
cfg = bench.bench_config(c["height"], c["width"], c["full_step"])
ops.set_conv_precision("bf16")
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to("cuda:0").train()
step = TrainStep(model, run_optimizers=c["full_step"], distributed=False)
batch = batch_to(make_stream_batch(c["batch"], c["windows"], c["height"], c["width"], 2, seed=0), "cuda:0")
batch["rng"] = {k: v.to("cuda:0") for k, v in make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0).items()}
for _ in range(2):
    step(batch)
ops._NC8_LOG = []
step(batch)
torch.cuda.synchronize()
log, ops._NC8_LOG = ops._NC8_LOG, None
by = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.defaultdict(lambda: collections.Counter())
for who, shp in log:
    n = 1
    for v in shp:
        n *= v
    by[who][0] += 1
    by[who][1] += 2.0 * n / 1e6
    shapes[who][shp] += 1
tot = sum(v[1] for v in by.values())
print(f"{len(log)} layout passes, {tot:.0f} MB of source tensors per step")
for who, (n, mb) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    top = sorted(shapes[who].items(), key=lambda kv: -kv[1] * __import__('math').prod(kv[0]))[:6]
    print(f"  {who:34s} {n:4d} passes {mb:8.0f} MB   " + "  ".join(f"{'x'.join(map(str, s))}*{k}" for s, k in top))
