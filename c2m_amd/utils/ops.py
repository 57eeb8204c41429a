"""Tensor ops of the reference's utils/ops.py that the hot path uses (ops.py:183-275), on HIP kernels.

`grid_sample` and `get_corresponding_map` (ops.py:183,205) are internal helpers of `resample` / `get_occlusion_map` in the
reference; they have no counterpart here (the sampling grid and the unclamped splat sum are never materialised) and,
with `c2m_amd.install_as_reference_layout()`, the reference's own definitions stay in place.

`resample` keeps the reference's coordinate quirk (align_corners=True grid sampled with align_corners=False, border
padding): zero flow is NOT the identity.  The grid is never materialised -- coordinates are computed in the kernel.
"""
import torch

from .. import ops as _ops


def resample(image, flow, mode='bilinear'):
    if mode != 'bilinear':
        raise NotImplementedError(mode)
    return _ops.flow_warp(image, flow)


def get_grid(batchsize, rows, cols, gpu_id=0, device=None):
    """Base grid in the align_corners=True convention (ops.py:196-202); device-agnostic (the reference hard-codes .cuda)."""
    device = device if device is not None else (torch.device("cuda", gpu_id) if torch.cuda.is_available() else "cpu")
    lx = torch.linspace(-1, 1, cols) if cols > 1 else torch.tensor([-1.0])
    ly = torch.linspace(-1, 1, rows) if rows > 1 else torch.tensor([-1.0])
    g = torch.stack([lx.view(1, cols).expand(rows, cols), ly.view(rows, 1).expand(rows, cols)], 0)
    return g.unsqueeze(0).repeat(batchsize, 1, 1, 1).to(device)


def mesh_grid(B, H, W):
    xs = torch.arange(0, W).repeat(B, H, 1)
    ys = torch.arange(0, H).repeat(B, W, 1).transpose(1, 2)
    return torch.stack([xs, ys], 1)


def get_occlusion_map(flow):
    occ, _ = _ops.occlusion_splat(flow, want_map=True)
    return occ
