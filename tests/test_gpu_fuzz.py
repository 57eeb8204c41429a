"""Random-geometry conv parity sweep (tools/fuzz_conv.py) as a test: forward, data gradient, weight / bias gradient of ops.conv
against torch's float64 CPU convolution on geometries the hand-written case lists do not enumerate, in both precision modes."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--seed", "11"], ["--seed", "12", "--bf16"], ["--seed", "23"]], ids=["fp32", "bf16", "fp32-seed23"])
def test_random_conv_geometries_vs_float64_reference(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_conv.py"), "--cases", "120"] + extra, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "0 failures" in out.stdout


def test_more_than_128_taps_is_refused_loudly():
    from c2m_amd import ops
    x = torch.randn(1, 4, 3, 8, 8, device="cuda:0")
    w = torch.randn(4, 4, 3, 7, 7, device="cuda:0")
    with pytest.raises(NotImplementedError, match="128 taps"):
        ops.conv(x, w, None, stride=1, padding=(1, 3, 3))


def test_random_norm_pool_resize_shapes_vs_float64_reference():
    """tools/fuzz_ops.py: norm + activation (BN / IN / SPADE, fwd + every gradient), x2 up-sampling, 2x2 max-pool, bilinear resize
    and masked L1 on odd extents around the kernels' chunk / vector / tile boundaries."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ops.py"), "--cases", "250", "--seed", "3"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "0 failures" in out.stdout


def test_random_shapes_through_the_forced_f4x4_winograd_kernel():
    """tools/fuzz_wino4.py: odd / large 2-D 3x3 geometries through conv_wino4.hip whatever the routing rule would pick (forward, data
    gradient incl. the two-target reflect form, lrelu epilogue, bit-repeatability) against float64 at the conv gates."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_wino4.py"), "--cases", "60", "--seed", "4"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "0 failures" in out.stdout
