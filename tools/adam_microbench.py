"""Multi-tensor Adam kernel on the bench model's parameter set: time per optimizer step and HBM GB/s
(28 B/element algorithmic: read p, g, m, v; write p, m, v), next to torch.optim.Adam(foreach) on the same tensors."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel

cfg = normalize_config(default_config(height=128, width=256, num_input_frames=2))
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to("cuda:0")
opts = [model.optimizer, model.optimizer_gnn, model.d_optimizer_image, model.d_optimizer_video]
params = [p for o in opts for p in o.param_groups[0]["params"]]
n = sum(p.numel() for p in params)
for p in params:
    p.grad = torch.randn_like(p) * 1e-3
ref = [torch.optim.Adam(o.param_groups[0]["params"], lr=1e-4, betas=(0.5, 0.999), eps=1e-7, foreach=True) for o in opts]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) * 1e3 / iters


g, w = timeit(lambda: [o.step() for o in opts])
print(f"c2m_amd Adam : {len(params)} tensors, {n / 1e6:.1f} M elements, {g:.3f} ms GPU / {w:.3f} ms wall per 4-optimizer step, "
      f"{28 * n / (g * 1e-3) / 1e9:.0f} GB/s algorithmic")
g, w = timeit(lambda: [o.step() for o in ref])
print(f"torch foreach: {g:.3f} ms GPU / {w:.3f} ms wall, {28 * n / (g * 1e-3) / 1e9:.0f} GB/s algorithmic")
