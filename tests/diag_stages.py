"""Diagnostic (not a test): stage-by-stage product (GPU) vs oracle (CPU) errors on an e2e fixture.
usage: python tests/diag_stages.py [fixture-name]"""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c2m_amd.config import normalize_config
from c2m_amd.modules.model import GeneratorFullModel, _stack_time
from c2m_amd.synthetic import make_batch, batch_to
from c2m_amd.modules.layers.common import fold_time, unfold_time
from oracle import c2m_oracle as O
from oracle.golden_util import synth_state
from golden_io import Case

name = sys.argv[1] if len(sys.argv) > 1 else "e2e_tin2_spade_full"
c = Case(name); m = c.meta
cfg = normalize_config(m["cfg"])
sd = synth_state(m["spec"], m["seed"])
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"], dataset="cityscapes")
model.load_state_dict(sd, strict=True); model.to("cuda:0").train()
batch = make_batch(m["batch_size"], 128, 256, m["t_in"], seed=m["seed"])
rng = c.group("rng"); rng["click_index"] = rng["click_index"].long()
gb = batch_to(batch, "cuda:0"); gb["rng"] = {k: v.to("cuda:0") for k, v in rng.items()}

def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return ((a - b).norm() / max(b.norm().item(), 1e-30)).item(), (a - b).abs().max().item(), b.abs().max().item()

S = O.State(sd)
t_in = m["t_in"]
ob = dict(batch); ob["tracking_gnn"] = batch["tracking_gnn"].clone()
# oracle stages
frames, bg, fg = ob["video"], ob["bg_mask"], ob["fg_mask"]
inst = ob["instance_mask"].float().int()
seg = torch.cat([bg[:, :, :t_in], fg[:, :, :t_in]], 1)
enc_in = torch.cat([O.stack_time_into_channels(frames[:, :, :t_in]), O.stack_time_into_channels(seg), O.stack_time_into_channels(inst[:, :, :t_in])], 1)
if ob.get("input_of") is not None:
    enc_in = torch.cat([enc_in, O.stack_time_into_channels(ob["input_of"][:, :, :t_in]), O.stack_time_into_channels(ob["input_occ"][:, :, :t_in])], 1)
oapp = O.appearance_encoder(S, cfg, enc_in, ob["tracking_gnn"])
mi = dict(frames=frames, bg_mask=bg, fg_mask=fg, instance=inst, target_bw_of=ob["target_bw_of"], target_bw_occ=ob["target_bw_occ"], tracking_gnn=ob["tracking_gnn"], latent=rng["latent_traj"])
oout = O.dense_motion_network(S, cfg, oapp, mi, rng)
# product stages
v = model._resize_inputs(gb.get)
papp = model.appearance_encoder({"first_frame": model._encoder_input(v), "tracking_gnn": gb["tracking_gnn"]})
for k in oapp:
    print(f"app {k:18s} relL2 %.2e maxabs %.2e scale %.2e" % err(papp[k], oapp[k]))
pm = dict(frames=v["frames"], bg_mask=v["bg_mask"], fg_mask=v["fg_mask"], instance=v["instance"], input_of=v["input_of"], input_occ=v["input_occ"],
          target_bw_of=gb["target_bw_of"], target_bw_occ=gb["target_bw_occ"], target_fw_of=None, target_fw_occ=None, tracking_gnn=gb["tracking_gnn"],
          latent=gb["rng"]["latent_traj"], eps=gb["rng"]["eps"], click_index=gb["rng"]["click_index"])
pout = model.motion_encoder(papp, pm)
for k in oout:
    if torch.is_tensor(oout[k]):
        print(f"motion {k:18s} relL2 %.2e maxabs %.2e scale %.2e" % err(pout[k], oout[k]))
T = 5
last = frames[:, :, t_in - 1]
ogen = O.generator(S, cfg, O.fold_time(last.unsqueeze(2).repeat(1, 1, T, 1, 1)), O.fold_time(oout["dense_motion_bw"]), O.fold_time(oout["occlusion_bw"]))
# feed the ORACLE's flow to the product generator to isolate it
rep = gb["video"][:, :, t_in - 1].unsqueeze(0).expand(T, *gb["video"][:, :, t_in - 1].shape).reshape(-1, 3, 128, 256)
pgen = model.generator(rep, fold_time(oout["dense_motion_bw"].detach().cuda()), fold_time(oout["occlusion_bw"].detach().cuda()))
print("generator(oracle flow)  relL2 %.2e maxabs %.2e scale %.2e" % err(pgen, ogen))
pgen2 = model.generator(rep, fold_time(pout["dense_motion_bw"]), fold_time(pout["occlusion_bw"]))
print("generator(product flow) relL2 %.2e maxabs %.2e scale %.2e" % err(pgen2, ogen))
