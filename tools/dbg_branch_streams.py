"""Which of the two branch-stream schemes (ops.aux_branch / ops.deferred_wgrads) changes a step, if any?  Runs the tiny model for a
few steps in each mode several times and reports the first step / gradients that differ from the all-off run.
   python tools/dbg_branch_streams.py [repeats=3] [steps=4]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd import ops
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep
from tests.test_gpu_optim import _tiny_cfg

DEV = torch.device("cuda:0")
repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BENCH = os.environ.get("DBG_BENCH_CONFIG")          # e.g. 1: the headline bench configuration instead of the tiny model
if BENCH is not None:
    import bench
    c = bench.CONFIGS[int(BENCH)]
    cfg = bench.bench_config(c["height"], c["width"], c["full_step"])
    ops.set_conv_precision("bf16" if c["dtype"] == "bf16" else "fp32")
else:
    cfg = _tiny_cfg()
tp = cfg["train_params"]


def run(aux, defer):
    ops._AUX, ops._DEFER_WGRAD = aux, defer
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=copy.deepcopy(tp), model_params=copy.deepcopy(cfg["model_params"]),
                               dataset="cityscapes").to(DEV).train()
    step = TrainStep(model, run_optimizers=True if BENCH is None else c["full_step"], distributed=False)
    out = []
    for it in range(steps):
        if BENCH is None:
            batch = batch_to(make_batch(2, 128, 256, 2, seed=60 + it), DEV)
            rng = make_step_rng(batch, z_dim=16, latent_dim=32, seed=it)
        else:
            from c2m_amd.synthetic import make_stream_batch
            batch = batch_to(make_stream_batch(c["batch"], c["windows"], c["height"], c["width"], 2, seed=it), DEV)
            rng = make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=it)
        batch["rng"] = {k: v.to(DEV) for k, v in rng.items()}
        _, lg, _ = step(batch)
        if os.environ.get("DBG_SYNC", "0") == "1":
            torch.cuda.synchronize()
        out.append((float(lg["total_gen"].detach()),
                    {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    return out


base = run("0", False)
for mode in (("0", False), ("1", False), ("0", True), ("1", True)):
    for r in range(repeats):
        got = run(*mode)
        msg = "same"
        for it, ((l0, g0), (l1, g1)) in enumerate(zip(base, got)):
            bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
            if bad or l0 != l1:
                msg = f"step {it}: loss {l0} vs {l1}; {len(bad)} gradients differ, last in module order: {bad[-4:]} first: {bad[:3]}"
                break
        print(f"aux={mode[0]} defer={mode[1]} run {r}: {msg}", flush=True)
