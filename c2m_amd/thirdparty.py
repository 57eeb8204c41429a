"""Product-side restatements of the two small third-party ops on the path (PyTorch device ops, negligible FLOPs --
SURVEY.md §2.4 marks them "support"): torchvision.ops.roi_align and torch_geometric.nn.GATv2Conv.
Neither package is a dependency; parameter names follow the originals so checkpoints load."""
import math

import torch
import torch.nn.functional as F
from torch import nn


def roi_align(inp, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """RoIAlign (aligned=False, adaptive sampling grid); inp [N,C,H,W], boxes [K,5] = (batch, x1, y1, x2, y2).
    The sampling-grid sizes need the box extents on the host: one small D2H copy of the [K,5] box table."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    _, C, H, W = inp.shape
    bx = boxes.detach().to("cpu", torch.float64)
    off = 0.5 if aligned else 0.0
    out = []
    for k in range(bx.shape[0]):
        b = int(bx[k, 0])
        x1, y1, x2, y2 = (float(bx[k, j]) * spatial_scale - off for j in (1, 2, 3, 4))
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        if gh <= 0 or gw <= 0:
            out.append(inp.new_zeros(C, ph, pw))
            continue
        dev, dt = inp.device, inp.dtype
        ys = (y1 + torch.arange(ph, device=dev, dtype=dt)[:, None] * bh +
              (torch.arange(gh, device=dev, dtype=dt)[None, :] + 0.5) * bh / gh).reshape(-1)
        xs = (x1 + torch.arange(pw, device=dev, dtype=dt)[:, None] * bw +
              (torch.arange(gw, device=dev, dtype=dt)[None, :] + 0.5) * bw / gw).reshape(-1)
        oy, ox = (ys < -1.0) | (ys > H), (xs < -1.0) | (xs > W)
        y, x = ys.clamp(min=0.0), xs.clamp(min=0.0)
        yl, xl = y.floor().long(), x.floor().long()
        ty, tx = yl >= H - 1, xl >= W - 1
        yh = torch.where(ty, torch.full_like(yl, H - 1), yl + 1)
        yl = torch.where(ty, torch.full_like(yl, H - 1), yl)
        y = torch.where(ty, yl.to(dt), y)
        xh = torch.where(tx, torch.full_like(xl, W - 1), xl + 1)
        xl = torch.where(tx, torch.full_like(xl, W - 1), xl)
        x = torch.where(tx, xl.to(dt), x)
        ly, lx = y - yl.to(dt), x - xl.to(dt)
        hy, hx = 1.0 - ly, 1.0 - lx
        f = inp[b]
        val = (hy[:, None] * hx[None, :]) * f[:, yl][:, :, xl] + (hy[:, None] * lx[None, :]) * f[:, yl][:, :, xh] + \
              (ly[:, None] * hx[None, :]) * f[:, yh][:, :, xl] + (ly[:, None] * lx[None, :]) * f[:, yh][:, :, xh]
        val = val * ((~oy)[:, None] & (~ox)[None, :]).to(dt)
        out.append(val.reshape(C, ph, gh, pw, gw).sum(dim=(2, 4)) / max(gh * gw, 1))
    return torch.stack(out, 0) if out else inp.new_zeros(0, C, ph, pw)


class GATv2Conv(nn.Module):
    """GATv2 (Brody et al.) with PyG's parameterisation: lin_l / lin_r (bias), att [1,H,C], bias [C];
    heads averaged (concat=False), no self loops, softmax over the incoming edges of each target node."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, add_self_loops=True, **_):
        super().__init__()
        if concat or add_self_loops:
            raise NotImplementedError("only concat=False, add_self_loops=False is on the C2M path")
        self.heads, self.out_channels, self.negative_slope = heads, out_channels, negative_slope
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, heads * out_channels, bias=True)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for w in (self.lin_l.weight, self.lin_r.weight, self.att):
            a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
            w.data.uniform_(-a, a)
        self.lin_l.bias.data.zero_()
        self.lin_r.bias.data.zero_()

    def forward(self, x, edge_index):
        N, H, C = x.shape[0], self.heads, self.out_channels
        xl = self.lin_l(x).view(N, H, C)
        xr = self.lin_r(x).view(N, H, C)
        src, dst = edge_index[0].long(), edge_index[1].long()
        e = F.leaky_relu(xl[src] + xr[dst], self.negative_slope)
        logit = (e * self.att).sum(-1)
        mx = torch.full((N, H), float("-inf"), dtype=x.dtype, device=x.device)
        mx = mx.scatter_reduce(0, dst[:, None].expand(-1, H), logit, reduce="amax", include_self=True)
        mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
        ex = (logit - mx[dst].detach()).exp()
        den = torch.zeros((N, H), dtype=x.dtype, device=x.device).index_add(0, dst, ex)
        alpha = ex / (den[dst] + 1e-16)
        out = torch.zeros((N, H, C), dtype=x.dtype, device=x.device).index_add(0, dst, xl[src] * alpha.unsqueeze(-1))
        return out.mean(dim=1) + self.bias
