"""Which model source lines issue the ATen (non-c2m) device kernels of a train step, and how many bytes they move?
A TorchDispatchMode over one step: per (aten op, innermost c2m_amd frame outside ops.py) -> calls, output MBytes.  Backward ops
run on the autograd thread without Python frames: they are listed under the op name alone.
   python tools/aten_sources.py [--config 1] [--top 60]"""
import argparse, collections, copy, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from c2m_amd import ops
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_stream_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--min-kb", type=float, default=256.0)
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
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
VIEW = ("view", "reshape", "_unsafe_view", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "detach",
        "alias", "as_strided", "t", "split", "split_with_sizes", "unbind", "chunk", "empty", "empty_like", "empty_strided",
        "record_stream", "is_same_size", "_local_scalar_dense", "stride", "sym_size", "unfold", "narrow", "new_empty", "lift_fresh")
stat = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEW:
            return out
        o = out[0] if isinstance(out, (tuple, list)) and out else out
        if not (torch.is_tensor(o) and o.is_cuda):
            return out
        kb = o.numel() * o.element_size() / 1024
        src = ""
        for fr in reversed(traceback.extract_stack(limit=14)):
            fn = fr.filename
            if "c2m_amd" in fn and not fn.endswith(("ops.py", "_lib.py")):
                src = f"{fn.split('c2m_amd/')[-1]}:{fr.lineno}"
                break
        else:
            for fr in reversed(traceback.extract_stack(limit=14)):
                if fr.filename.endswith("ops.py"):
                    src = f"ops.py:{fr.lineno} ({fr.name})"
                    break
        if kb >= 4096:
            shp = [tuple(t.shape) if torch.is_tensor(t) else "." for t in args[:3]]
            contig = [t.is_contiguous() if torch.is_tensor(t) else "." for t in args[:3]]
            shapes[(name, src, str(shp), str(contig))] += kb / 1024
        e = stat[(name, src, "big" if kb >= a.min_kb else "small")]
        e[0] += 1; e[1] += kb / 1024
        return out


with Spy():
    step(batch)
torch.cuda.synchronize()
tot = collections.Counter(); cnt = collections.Counter()
for (name, src, size), (n, mb) in stat.items():
    tot[size] += mb; cnt[size] += n
print(f"ATen device ops in one step: {cnt['big']} with >= {a.min_kb:.0f} KB outputs ({tot['big']:.0f} MB written), {cnt['small']} smaller")
for (name, src, size), (n, mb) in sorted(stat.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{mb:9.1f} MB {n:5d}x  {name:28s} {src or '(autograd thread)'}")
print("-- single big ops (>= 4 MB out): MB, op, source, arg shapes, contiguous?")
for (name, src, shp, contig), mb in shapes.most_common(70):
    print(f"{mb:8.1f} MB  {name:12s} {src or '(autograd thread)':60s} {shp} {contig}")
