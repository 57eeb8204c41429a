"""Diagnostic (not a test; CPU only, ~1 min): how far the fp32 oracle itself is from the same graph in fp64, per
parameter gradient, on the full-width B=1 step.  Result (round 1): median 6.6e-4; the deepest keys of the backward pass
(generator.flowembedder.conv_first.*, generator.middle.3.*) 2e-3..4e-3 -- the step functions (ReLU / LeakyReLU / max-pool
/ L1 sign) make fp32 gradients this sensitive to rounding, so a fp32 implementation cannot be held tighter than ~1e-2
against another fp32 implementation on those keys (tests/test_gpu_model.py::test_full_width_step_vs_oracle)."""
import sys, copy, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng
from oracle import c2m_oracle as O
cfg = normalize_config(default_config(num_input_frames=2, use_image_discriminator=False, use_video_discriminator=False))
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
sd = {k: v.clone() for k, v in model.state_dict().items()}
del model
batch = make_batch(1, 128, 256, 2, seed=0)
rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
_orig = O.generate_sparse_motion
def _wrap(cfg, gnn, thetas, inst, use_gt):
    g = gnn.clone()
    g.targets_theta = g.targets_theta.float()
    out = _orig(cfg, g, {k: v.detach().float() for k, v in thetas.items()}, inst.float(), use_gt)
    return {k: (v.to(CUR[0]) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in out.items()}
O.generate_sparse_motion = _wrap
CUR = [torch.float32]
def run(dt):
    CUR[0] = dt
    S = O.State({k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()})
    b = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    g = batch["tracking_gnn"].clone()
    for a in ("x", "targets_theta", "source_frames_nodes_roi_padded"):
        if hasattr(g, a) and getattr(g, a).is_floating_point(): setattr(g, a, getattr(g, a).to(dt))
    b["tracking_gnn"] = g
    r = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in rng.items()}
    t0 = time.time()
    oo, olg, _, _ = O.forward(S, cfg, b, r)
    O.train_step_backward(cfg, olg, {}, {})
    print(dt, "time", time.time() - t0, {k: float(v) for k, v in list(olg.items())[:4]})
    return S.grads()
g32 = run(torch.float32)
g64 = run(torch.float64)
rel = []
for k in g32:
    a, b = g32[k].double(), g64[k].double()
    rel.append(((a - b).norm().item() / max(b.norm().item(), 1e-30), k))
rel.sort(reverse=True)
print("median", rel[len(rel)//2][0])
for e, k in rel:
    if 'flowembedder.conv_first' in k or 'flowembedder.down_blocks.0' in k or 'generator.middle.3' in k: print(f"{e:.2e} {k}")
print("top (non-bias):", [(f"{e:.2e}", k) for e, k in rel if not k.endswith('conv.bias')][:8])
