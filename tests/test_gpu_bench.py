"""bench.py contract checks on the GPU box (one JSON line, required keys; torchrun plumbing with the RCCL reducer path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_contract():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _json_line(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["value"] > 0 and r["roofline"]["bound"] == "mfma"
    rf = r["roofline"]
    assert 0 < rf["frac"] < 1 and 0 < rf["whole_step"]["frac"] < 1
    assert rf["achieved"] <= rf["algorithmic_tflops"]              # executed MFMA FLOPs never exceed the algorithmic count
    for fam in rf["families"].values():
        assert 0 < fam["frac"] < 1, fam                             # every family is priced against the MFMA pipe it runs on
    assert r["rccl_ranks"] == 0 and "config" in r and "BASELINE configs[1]" in r["config"]["workload"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus 2 on a one-GPU box must fail loudly (never a silent 1-rank line), with or without a launcher."""
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_under_torchrun_with_rccl_reducer():
    """1 rank under torch.distributed.run, all-reduce path forced: buckets, side stream, RCCL, stream join."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch",
           "1", "--no-cpu-baseline", "--no-roofline", "--force-reducer", "--measure-comm"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["rccl_ranks"] == 1
    assert r["allreduce_bytes_per_step"] > 4e8 and r["allreduce_buckets"] >= 10
    assert r["allreduce_exposed_ms_per_step"] is not None
    pb = r["allreduce_exposed_ms_per_bucket"]         # per-bucket share of the exposed time, in launch order
    assert pb is not None and len(pb) == r["allreduce_buckets"] and all(v >= 0 for _, v in pb)
    assert abs(sum(v for _, v in pb) - r["allreduce_exposed_ms_per_step"]) < 0.5


def test_two_rank_rehearsal_on_one_gpu_gradients_identical():
    """Two ranks under torch.distributed.run sharing the box's single GPU (collectives over gloo, because RCCL cannot put
    two ranks on one device): different data per rank, bucketed all-reduce overlapped with backward on HIP tensors, and
    every rank must end up with bit-identical gradients.  The production launch uses RCCL (previous test)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C2M_REHEARSAL_SHARED_GPU="1")
    env.pop("WORLD_SIZE", None)
    # plain `python bench.py --gpus 2`: bench.py starts the two ranks itself (the form the scaling driver may use)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "2", "--batch", "1", "--no-cpu-baseline",
           "--no-roofline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["n_gpus"] == 2 and r["value"] > 0 and "bit-identical" in r["grad_sync"] and "2 rank" in r["grad_sync"]
    assert r["allreduce_bytes_per_step"] > 4e8
    assert r["config"]["global_batch"] == 2 and r["config"]["parallelism"] == "dp2"


def test_two_rank_rccl_bench_when_two_gpus_are_visible():
    """Any lease with >= 2 GPUs produces an N = 2 RCCL line by itself: `python bench.py --gpus 2` (self-spawned ranks, one
    per GPU, backend nccl = RCCL over xGMI), mean-of-ranks gradients bit-identical on both ranks, the exposed all-reduce
    time measured.  The reference's launch: src/train.py:57-60,144-159.  Skipped on one-GPU boxes (the gloo rehearsal
    above covers the reducer logic there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "C2M_REHEARSAL_SHARED_GPU"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                          "--no-roofline"], cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["value"] > 0
    assert "bit-identical" in r["grad_sync"] and "2 rank" in r["grad_sync"]
    assert r["allreduce_exposed_ms_per_step"] is not None and r["allreduce_bytes_per_step"] > 4e8
    assert r["config"]["global_batch"] == 16 and r["config"]["parallelism"] == "dp2"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_2rank_bench.json"), "w") as f:      # evidence for profiles/
        json.dump(r, f)


@pytest.mark.parametrize("config,expect", [(2, ("256x512", "bf16", "full adversarial")), (3, ("128x256", "bf16", "full adversarial")),
                                           (4, ("256x512", "2 windows", "full adversarial"))])
def test_bench_baseline_config_presets(config, expect):
    """`--config K` runs BASELINE.json configs[K] (here at batch 1 to stay short): resolution, bf16 conv mode, both
    discriminators + the four Adam steps, and for configs[4] two 7-frame windows per stream sample."""
    out = subprocess.run([sys.executable, "bench.py", "--config", str(config), "--batch", "1", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _json_line(out.stdout)
    assert r["dtype"] == "bf16" and r["value"] > 0 and f"configs[{config}]" in r["config"]["workload"]
    for frag in expect:
        assert frag in r["config"]["workload"], (frag, r["config"]["workload"])
    assert r["config"]["global_batch"] == (2 if config == 4 else 1)
    assert r["roofline"]["peak"] == 2500.0 and 0 < r["roofline"]["frac"] < 1
    assert r["hip_graph"] is False and r["hip_graph_replay"]["value"] > 0       # the eager line + the replay side object


def test_bench_graph_side_measurement():
    out = subprocess.run([sys.executable, "bench.py", "--graph", "--batch", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _json_line(out.stdout)
    assert r["hip_graph"] is True and r["value"] > 0 and "roofline" not in r


def test_bench_graph_replay_with_the_reducer():
    """HIP-graph replay of zero_grad + forward + backward with the gradient reducer in place: gradients accumulate into the
    reducer's static buckets inside the graph, the RCCL all-reduces run eagerly after each replay (1 rank, forced)."""
    out = subprocess.run([sys.executable, "bench.py", "--force-reducer", "--graph", "--batch", "1", "--steps", "2", "--warmup", "2",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["hip_graph"] is True and r["rccl_ranks"] == 1 and r["value"] > 0 and r["allreduce_buckets"] >= 10
    assert r["allreduce_op"] == "avg"


def test_bench_side_configs_object():
    """Round 5: the default one-GPU line carries `side_configs` -- the bf16 BASELINE configurations as HIP-graph replays measured in
    the same process after (and outside) the headline timed region.  Here: configs[3] only, two replays, on a short headline run."""
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline",
                          "--no-roofline", "--side-configs", "3", "--side-steps", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _json_line(out.stdout)
    sc = r["side_configs"]["configs[3]"]
    assert "error" not in sc, sc
    assert sc["steps"] == 2 and sc["ms_per_step"] > 0 and sc["unit"] == "frames/s" and "BASELINE configs[3]" in sc["workload"]
    assert abs(sc["value"] - 8 * 7 * 1000.0 / sc["ms_per_step"]) < 0.02 * sc["value"]
    assert 0 < sc["conv_frac_of_peak"] < 1 and sc["peak_tflops"] == 2500.0 and sc["conv_launches_per_step"] > 100
    assert r["config"]["workload"].startswith("BASELINE configs[1]") and r["dtype"] == "f32"      # the headline is untouched
