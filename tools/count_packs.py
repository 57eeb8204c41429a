import os, sys, collections, copy
sys.path.insert(0, "/root/repo")
import torch, bench
from c2m_amd import ops
from c2m_amd.train import TrainStep
from c2m_amd.modules.model import GeneratorFullModel
dev = torch.device("cuda", 0)
cfg = bench.bench_config(128, 256, False)
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes").to(dev).train()
step = TrainStep(model, run_optimizers=False, distributed=False)
batch = bench.batch_to(bench.make_stream_batch(8, 1, 128, 256, 2, seed=0), dev)
rng = bench.make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
batch["rng"] = {k: v.to(dev) for k, v in rng.items()}
step(batch); torch.cuda.synchronize()
cnt = collections.Counter(); frozen = collections.Counter()
orig = ops._packed
def traced(w, fr, kind, build):
    (frozen if fr else cnt)[(id(w), str(kind))] += 1
    return orig(w, fr, kind, build)
ops._packed = traced
step(batch); torch.cuda.synchronize()
print("non-frozen pack calls", sum(cnt.values()), "unique (weight, kind)", len(cnt), "frozen calls", sum(frozen.values()))
print(collections.Counter(cnt.values()))
