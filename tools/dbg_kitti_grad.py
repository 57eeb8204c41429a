"""Debug (GPU box): product vs oracle input gradients of the kitti + SPADE generator fixture, max-abs and norm-wise, and where they differ."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from golden_io import Case
from oracle import c2m_oracle as O
from oracle.golden_util import synth_state, synth_input
from c2m_amd.modules.generator.generator import OcclusionAwareGenerator

name = sys.argv[1] if len(sys.argv) > 1 else "mod_generator_kitti_spade"
c = Case(name)
m = c.meta
seed = m["seed"]
rnd = lambda s, *sh: torch.randn(tuple(sh), generator=torch.Generator().manual_seed(s))
DEV = torch.device("cuda", 0)
mod = OcclusionAwareGenerator(copy.deepcopy(m["generator"]), copy.deepcopy(m["flow_embedder"]), input_channel=3, dataset=m.get("dataset", "cityscapes"))
mod.load_state_dict(synth_state(m["spec"], seed), strict=True)
mod.to(DEV).train()
inp = {k: synth_input(v).to(DEV).requires_grad_(True) for k, v in m["inputs"].items()}
y = mod(inp["first_frame"], inp["flow"], inp["occlusion_map"])
(y * rnd(seed + 100, *y.shape).to(DEV)).sum().backward()
S = O.State(synth_state(m["spec"], seed))
oi = {k: synth_input(v).requires_grad_(True) for k, v in m["inputs"].items()}
cfg = {"model_params": {"generator": m["generator"], "flow_embedder": m["flow_embedder"]}}
yo = O.generator(S, cfg, oi["first_frame"], oi["flow"], oi["occlusion_map"], p="", dataset=m.get("dataset", "cityscapes"))
(yo * rnd(seed + 100, *yo.shape)).sum().backward()
print("out max abs", float((y.detach().cpu() - yo.detach()).abs().max()), "scale", float(yo.abs().max()))
for k in inp:
    a, b = inp[k].grad.cpu().double(), oi[k].grad.double()
    d = (a - b).abs()
    sc = float(b.abs().max())
    print(f"d{k}: max abs {float(d.max()):.3e} / scale {sc:.3e} = {float(d.max()) / sc:.2e};  L2 rel {float((a - b).norm() / b.norm()):.2e};  "
          f"elements above 5e-3*scale: {int((d > 5e-3 * sc).sum())} of {d.numel()};  argmax {np.unravel_index(int(d.argmax()), d.shape)}")
og = S.grads()
gg = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
rows = []
for k in og:
    a, b = gg[k].cpu().double(), og[k].double()
    rows.append((float((a - b).norm() / max(float(b.norm()), 1e-30)), k))
rows.sort(reverse=True)
print("worst parameter gradients (L2 rel):", [(f"{e:.1e}", k) for e, k in rows[:10]])
