"""Which normalisation calls a bench configuration makes: (mode, N, C, S, affine / SPADE, act, requires_grad) with counts.
    python tools/norm_shapes.py [--config 1]
"""
import argparse, collections, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from c2m_amd import ops
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_stream_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
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
step(batch)
seen = collections.Counter()
orig = ops._NormActFn.apply


def spy(x, gamma, beta, gb, rm, rv, mode, act, eps, mom, *rest):      # *rest: feeds, private_input
    S = 1
    for d in x.shape[2:]:
        S *= d
    kind = "spade" if gb is not None else ("affine" if gamma is not None else "plain")
    seen[(mode, x.shape[0], x.shape[1], S, kind, act, bool(x.requires_grad))] += 1
    return orig(x, gamma, beta, gb, rm, rv, mode, act, eps, mom, *rest)


ops._NormActFn.apply = spy
step(batch)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] * kv[1]):
    print("mode %d N %3d C %4d S %7d %-6s act %-5s grad %d  x%d   (%.1f MB)" % (k + (v, k[1] * k[2] * k[3] * 4 / 1e6)))
