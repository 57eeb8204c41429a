"""Which ops receive non-contiguous tensors (each costs a torch copy kernel)?  One bench step with ops._f instrumented, plus
torch.profiler's view of the copy / add kernels with their Python call sites.  usage: python tools/trace_copies.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy, torch
import bench
from c2m_amd import ops
from c2m_amd.train import TrainStep
from c2m_amd.modules.model import GeneratorFullModel  # noqa

dev = torch.device("cuda", 0)
cfg = bench.bench_config(128, 256, False)
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to(dev).train()
step = TrainStep(model, run_optimizers=False, distributed=False)
batch = bench.batch_to(bench.make_stream_batch(8, 1, 128, 256, 2, seed=0), dev)
rng = bench.make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0)
batch["rng"] = {k: v.to(dev) for k, v in rng.items()}
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
stat = collections.Counter(); byt = collections.Counter()
orig = ops._f
def traced(t):
    if not t.is_contiguous():
        st = traceback.extract_stack(limit=6)
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st[:-1]))
        stat[key] += 1; byt[key] += t.numel() * 4
    return orig(t)
ops._f = traced
step(batch)
torch.cuda.synchronize()
ops._f = orig
print("non-contiguous inputs to ops (per step):", sum(stat.values()), "copies,", sum(byt.values()) / 1e6, "MB")
for k, n in byt.most_common(25):
    print(f"{stat[k]:4d} x {n / 1e6:8.1f} MB  {k}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(batch)
    torch.cuda.synchronize()
# who calls the small torch kernels?  walk the event tree: for every aten op of interest, the nearest enclosing non-aten parent
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::clone", "aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::contiguous", "aten::mul", "aten::fill_",
                   "aten::zero_", "aten::index", "aten::sum", "aten::_to_copy"):
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name[:48])
            par = par.cpu_parent
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and ev.cpu_parent.name in (
                "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::to", "aten::zero_"):
            continue        # counted at the parent
        shp = str(ev.input_shapes[:1])[:40] if ev.input_shapes else ""
        k = (ev.name, " <- ".join(chain))
        agg[k][0] += 1
        agg[k][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("torch-level small ops by caller (GPU us per step):")
for (name, chain), (n, us) in rows[:40]:
    print(f"{us:9.1f} us {n:4d} x  {name:18s} <- {chain}")
