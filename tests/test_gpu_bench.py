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
    assert 0 < r["roofline"]["frac"] < 1


def test_bench_under_torchrun_with_rccl_reducer():
    """1 rank under torch.distributed.run, all-reduce path forced: buckets, side stream, RCCL, stream join."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch",
           "1", "--no-cpu-baseline", "--no-roofline", "--force-reducer"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["n_gpus"] == 1 and r["value"] > 0


def test_two_rank_rehearsal_on_one_gpu_gradients_identical():
    """Two ranks under torch.distributed.run sharing the box's single GPU (collectives over gloo, because RCCL cannot put
    two ranks on one device): different data per rank, bucketed all-reduce overlapped with backward on HIP tensors, and
    every rank must end up with bit-identical gradients.  The production launch uses RCCL (previous test)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C2M_REHEARSAL_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "2", "--batch",
           "1", "--no-cpu-baseline", "--no-roofline", "--check-grads"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    r = _json_line(out.stdout)
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["grad_sync"] == "identical on all ranks"
    assert r["config"]["global_batch"] == 2 and r["config"]["parallelism"] == "dp2"
