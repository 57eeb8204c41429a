"""Host-side checks of the Winograd F(4x4,3x3) path (no GPU): the transform matrices the kernel hard-codes satisfy the Toom-Cook
identity exactly, their B^T / A^T entries are dyadic (exact in fp32), a float64 emulation reproduces a direct convolution, and the
routing rule sends the large-grid 2-D layers -- and only those, fp32 only -- to conv_wino4.hip."""
import os
import re
import sys
from fractions import Fraction as Fr

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from wino43_matrices import matrices  # noqa: E402


def test_matrices_satisfy_the_winograd_identity_exactly():
    AT, G, BT = matrices()
    for i in range(4):
        for k in range(3):
            for l in range(6):
                s = sum(AT[i][j] * G[j][k] * BT[j][l] for j in range(6))
                assert s == (1 if l == i + k else 0), (i, k, l, s)
    dyadic = lambda v: v.denominator & (v.denominator - 1) == 0
    assert all(dyadic(v) for row in BT for v in row) and all(dyadic(v) for row in AT for v in row)


def test_kernel_source_holds_these_coefficients():
    """conv_wino4.hip spells the matrices out as float literals: every non-trivial |entry| of B^T and A^T must appear in it."""
    AT, G, BT = matrices()
    src = open(os.path.join(ROOT, "c2m_amd", "csrc", "conv_wino4.hip")).read()
    lits = {float(m) for m in re.findall(r"(?<![\w.])(\d+\.\d+)f", src)}
    need = {abs(float(v)) for M in (AT, BT) for row in M for v in row} - {0.0, 1.0}
    assert need <= lits, sorted(need - lits)
    for num, den in ((64, 81), (128, 243), (32, 81), (8, 27), (32, 243), (16, 81)):        # G, written as fractions
        assert f"({num}.f / {den}.f)" in src


def test_float64_emulation_matches_direct_convolution():
    AT, G, BT = (torch.tensor([[float(v) for v in row] for row in M], dtype=torch.float64) for M in matrices())
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 8, 12, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.float64)
    tiles = F.pad(x, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)
    U = torch.einsum("ij,mcjk,lk->mcil", G, w, G)
    Y = torch.einsum("ij,nmtujk,lk->nmtuil", AT, torch.einsum("mcil,nctuil->nmtuil", U, V), AT)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(2, 7, 8, 12)
    torch.testing.assert_close(y, F.conv2d(x, w, padding=1), rtol=1e-12, atol=1e-12)


def test_routing_rule():
    from c2m_amd import ops
    dev = torch.device("cpu")
    plan = lambda xs, cout, bf16=False, reflect=True: ops._ConvPlan(xs, (cout, xs[1], 3, 3), (1, 1, 1), (0, 1, 1), reflect, dev, bf16, None)
    big = plan((40, 128, 64, 128), 128)                     # 1 280 workgroups of 16x32 regions
    assert big.wino_fwd and big.wino4_fwd
    assert not plan((40, 256, 16, 32), 256).wino4_fwd       # 160 workgroups: stays on F(2x2,3x3)
    assert not plan((40, 128, 32, 64), 128).wino4_fwd       # 320 workgroups
    assert not plan((40, 32, 128, 256), 64).wino4_fwd       # K < 64
    assert not plan((40, 128, 64, 128), 96).wino4_fwd       # M not a multiple of 64
    assert not plan((40, 128, 64, 128), 128, bf16=True).wino4_fwd
    # reflect data gradient (round 5): the exact 64 x 128 domain on F(4x4) + the pad ring (conv_ring.hip), not the padded 66 x 130
    # domain (67 % fill of its regions), which stays the C2M_RING=off route
    assert big.wino4_dgrad and big.ring_dgrad
    assert plan((40, 128, 64, 128), 128, reflect=False).wino4_dgrad
    small = plan((40, 128, 32, 64), 128)                    # <= 32 x 64 maps: F(2x2) over the exact domain + ring
    assert small.ring_dgrad and not small.wino4_dgrad
    wide = plan((40, 32, 128, 256), 32)                     # F(2x2) on >= 64 x 128 maps: the padded domain (fill 0.8-0.9) stays
    assert wide.wino_dgrad and not wide.ring_dgrad
    assert not plan((40, 256, 16, 32), 256).ring_dgrad      # 16 x 32 maps (< 1024 pixels): the ring launch costs what the domain saves
    monkey = ops._RING
    try:
        ops._RING = "off"
        off = plan((40, 128, 64, 128), 128)
        assert off.wino_dgrad and not off.ring_dgrad and not off.wino4_dgrad
    finally:
        ops._RING = monkey
