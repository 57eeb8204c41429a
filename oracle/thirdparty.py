"""TEST INFRASTRUCTURE -- not product code.

Restatements of the three third-party algorithms the C2M hot path calls but which are
NOT vendored under /root/reference and are not installed in this image:

  * torchvision.ops.roi_align      (reference call site: src/modules/appearance_encoder/appearance_encoder.py:67-69)
  * torch_geometric.nn.GATv2Conv   (reference call site: src/modules/motion_estimator/sparse_motion_estimator.py:115-116)
  * torchvision.models.vgg19       (reference call site: src/modules/layers/vgg.py:13, slices vgg.py:35-81)

Neither dependency is version-pinned by the reference (README.md:23 points at a c2m.yml that is
absent), so parity for these three is "unpinned": we restate the published algorithms
(torchvision roi_align with aligned=False / sampling_ratio=-1, GATv2 from Brody et al. as
implemented by PyG's GATv2Conv with share_weights=False, the VGG-19 "E" configuration) and anchor
on the reference's own call sites.  Everything else on the path is pinned by golden vectors
captured from the live reference (oracle/capture_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math

import torch
import torch.nn.functional as F

# VGG-19 ("E") feature stack; integers are conv3x3 output widths, "M" is maxpool 2x2.
# torchvision `features` indices: conv at 0,2,5,7,10,12,14,16,19,21,23,25,28,30,32,34.
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19_feature_layout():
    """Yield (features_index, kind, cin, cout) for torchvision's vgg19().features[0:36]."""
    out, idx, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            out.append((idx, "pool", cin, cin))
            idx += 1
        else:
            out.append((idx, "conv", cin, v))
            out.append((idx + 1, "relu", v, v))
            idx += 2
            cin = v
    return out


def _bilinear_samples(feat, ys, xs):
    """torchvision roi_align `bilinear_interpolate` for a [C,H,W] map at sample grids ys[Sy], xs[Sx].

    Returns [C,Sy,Sx].  Samples with y<-1, y>H, x<-1 or x>W contribute 0.
    """
    C, H, W = feat.shape
    oob_y = (ys < -1.0) | (ys > H)
    oob_x = (xs < -1.0) | (xs > W)
    y = ys.clamp(min=0.0)
    x = xs.clamp(min=0.0)
    y_low = y.floor().long()
    x_low = x.floor().long()
    top = y_low >= H - 1
    y_high = torch.where(top, torch.full_like(y_low, H - 1), y_low + 1)
    y_low = torch.where(top, torch.full_like(y_low, H - 1), y_low)
    y = torch.where(top, y_low.to(y.dtype), y)
    right = x_low >= W - 1
    x_high = torch.where(right, torch.full_like(x_low, W - 1), x_low + 1)
    x_low = torch.where(right, torch.full_like(x_low, W - 1), x_low)
    x = torch.where(right, x_low.to(x.dtype), x)
    ly = y - y_low.to(y.dtype)
    lx = x - x_low.to(x.dtype)
    hy = 1.0 - ly
    hx = 1.0 - lx
    v1 = feat[:, y_low][:, :, x_low]
    v2 = feat[:, y_low][:, :, x_high]
    v3 = feat[:, y_high][:, :, x_low]
    v4 = feat[:, y_high][:, :, x_high]
    w1 = hy[:, None] * hx[None, :]
    w2 = hy[:, None] * lx[None, :]
    w3 = ly[:, None] * hx[None, :]
    w4 = ly[:, None] * lx[None, :]
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    keep = (~oob_y)[:, None] & (~oob_x)[None, :]
    return val * keep.to(val.dtype)


def roi_align(inp, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """RoIAlign (He et al.) as torchvision.ops.roi_align implements it.

    inp [N,C,H,W]; boxes [K,5] = (batch_index, x1, y1, x2, y2); returns [K,C,ph,pw].
    """
    if isinstance(output_size, int):
        ph = pw = output_size
    else:
        ph, pw = output_size
    K = boxes.shape[0]
    C = inp.shape[1]
    out = []
    offset = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(boxes[k, 0].item())
        x1 = float(boxes[k, 1]) * spatial_scale - offset
        y1 = float(boxes[k, 2]) * spatial_scale - offset
        x2 = float(boxes[k, 3]) * spatial_scale - offset
        y2 = float(boxes[k, 4]) * spatial_scale - offset
        roi_w = x2 - x1
        roi_h = y2 - y1
        if not aligned:
            roi_w = max(roi_w, 1.0)
            roi_h = max(roi_h, 1.0)
        bin_h = roi_h / ph
        bin_w = roi_w / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pw))
        gh = max(gh, 0)
        gw = max(gw, 0)
        count = max(gh * gw, 1)
        if gh == 0 or gw == 0:
            out.append(inp.new_zeros(C, ph, pw))
            continue
        iy = torch.arange(gh, dtype=inp.dtype, device=inp.device)
        ix = torch.arange(gw, dtype=inp.dtype, device=inp.device)
        py = torch.arange(ph, dtype=inp.dtype, device=inp.device)
        px = torch.arange(pw, dtype=inp.dtype, device=inp.device)
        ys = (y1 + py[:, None] * bin_h + (iy[None, :] + 0.5) * bin_h / gh).reshape(-1)
        xs = (x1 + px[:, None] * bin_w + (ix[None, :] + 0.5) * bin_w / gw).reshape(-1)
        vals = _bilinear_samples(inp[b], ys, xs)  # [C, ph*gh, pw*gw]
        vals = vals.reshape(C, ph, gh, pw, gw).sum(dim=(2, 4)) / count
        out.append(vals)
    if not out:
        return inp.new_zeros(0, C, ph, pw)
    return torch.stack(out, 0)


def gatv2_conv(x, edge_index, lin_l_w, lin_l_b, lin_r_w, lin_r_b, att, bias, heads, negative_slope=0.2):
    """GATv2 layer (Brody et al. 2022) as PyG's GATv2Conv(heads=H, concat=False, add_self_loops=False,
    share_weights=False) computes it.

    x [N,Fin]; edge_index [2,E] (row 0 = source j, row 1 = target i); att [1,H,C]; returns [N,C].
        e_ij = att . LeakyReLU(W_l x_j + W_r x_i);  alpha = softmax_j(e_ij) over incoming edges of i
        out_i = mean_h( sum_j alpha_ij W_l x_j ) + bias
    """
    N = x.shape[0]
    H = heads
    C = att.shape[-1]
    xl = F.linear(x, lin_l_w, lin_l_b).view(N, H, C)
    xr = F.linear(x, lin_r_w, lin_r_b).view(N, H, C)
    src, dst = edge_index[0].long(), edge_index[1].long()
    e = F.leaky_relu(xl[src] + xr[dst], negative_slope)
    logit = (e * att).sum(-1)  # [E,H]
    # softmax over edges sharing a target node
    mx = torch.full((N, H), float("-inf"), dtype=x.dtype, device=x.device)
    mx = mx.scatter_reduce(0, dst[:, None].expand(-1, H), logit, reduce="amax", include_self=True)
    mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    ex = (logit - mx[dst].detach()).exp()
    den = torch.zeros((N, H), dtype=x.dtype, device=x.device).index_add(0, dst, ex)
    alpha = ex / (den[dst] + 1e-16)
    msg = xl[src] * alpha.unsqueeze(-1)
    out = torch.zeros((N, H, C), dtype=x.dtype, device=x.device).index_add(0, dst, msg)
    return out.mean(dim=1) + bias


# ------------------------------------------------------------------------------------------------ FlowNet2 operators
# Python faces of the C restatements in oracle/c2m_oracle_index.c (oc_resample2d / oc_channelnorm / oc_correlation) of the
# reference's CUDA extensions src/modules/third_party/{resample2d,channelnorm,correlation} -- "parity unpinned": the
# extensions cannot be built here and the reference holds no fixtures for them.  CPU float32 tensors in, tensors out.
def _oc():
    import ctypes
    from . import build as _build
    L = _build.load()
    L.oc_correlation_out_size.restype = ctypes.c_int
    return L, ctypes


def _fp(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def resample2d(img, flow):
    L, ct = _oc()
    img, flow = img.detach().contiguous().float(), flow.detach().contiguous().float()
    N, C, H, W = img.shape
    out = torch.empty_like(img)
    L.oc_resample2d(_fp(img), _fp(flow), _fp(out), N, C, H, W)
    return out


def channelnorm(x):
    L, ct = _oc()
    x = x.detach().contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty(N, 1, H, W)
    L.oc_channelnorm(_fp(x), _fp(out), N, C, ct.c_long(H * W))
    return out


def correlation(a, b, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    L, ct = _oc()
    a, b = a.detach().contiguous().float(), b.detach().contiguous().float()
    N, C, H, W = a.shape
    D = 2 * (max_displacement // stride2) + 1
    oH = L.oc_correlation_out_size(H, pad_size, kernel_size, max_displacement, stride1)
    oW = L.oc_correlation_out_size(W, pad_size, kernel_size, max_displacement, stride1)
    out = torch.empty(N, D * D, oH, oW)
    L.oc_correlation(_fp(a), _fp(b), _fp(out), N, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    return out
