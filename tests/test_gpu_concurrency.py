"""Kernel concurrency under test: every kernel family on one stream while a foreign kernel stream
runs on another -- the bf16 gather kernel (the neighbour of round 3's wrong-sums finding), the bf16 patch / wide-wgrad kernels,
the fp32 Winograd kernels with their hand-written v_pk_add_f32, a reduce-copy shaped bandwidth stream, and RCCL's own all-reduce
kernel on a 1-rank "nccl" group (what the gradient reducer's side stream runs at N > 1).  Outputs must be bit-identical to the
solo run in both launch orders.  tools/concurrency_stress.py holds the victims / neighbours (also a CLI for tuning builds)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

VICTIMS = ["direct_s2", "direct_3d_s2", "patch3x3", "thin7x7", "thin3x3", "wino2_deep", "wino2_32rows", "wino4_zeros",
           "wino4_reflect", "wino3d", "nc8_3x3", "nc8_s2", "norm", "warp", "glue", "adam"]
NEIGHBOURS = ["bf16_igemm", "bf16_patch", "fp32_wino", "bandwidth"]


@pytest.fixture(scope="module")
def stress():
    import concurrency_stress as cs
    vic = cs.make_victims()
    assert sorted(vic) == sorted(VICTIMS)
    agg = cs.make_aggressors(NEIGHBOURS)
    yield cs, vic, agg
    del vic, agg
    torch.cuda.empty_cache()


@pytest.mark.parametrize("victim", VICTIMS)
def test_kernel_family_is_bit_identical_next_to_foreign_kernels(stress, victim):
    cs, vic, agg = stress
    for name in NEIGHBOURS:
        bad = cs.run_pair(vic[victim], agg[name], reps=2, burst=10)
        assert not bad, f"{victim} next to {name}: (rep, order, output, differing elements, max |diff|) {bad[:8]}"


def test_kernel_families_next_to_rccl_allreduce():
    """The same victims next to dist.all_reduce on a 1-rank nccl (= RCCL) group, in its own process (the process group must not
    leak into the other test modules)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "concurrency_stress.py"), "--aggressors", "rccl", "--reps", "2"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "outputs that differed: 0" in r.stdout and r.stdout.count("bit-identical") == len(VICTIMS)
