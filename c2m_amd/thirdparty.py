"""Product-side restatement of torch_geometric.nn.GATv2Conv (PyTorch device ops on a 24-node graph, negligible FLOPs --
SURVEY.md §2.4 marks it "support").  torch_geometric is not a dependency; parameter names follow the original so
checkpoints load.  (torchvision.ops.roi_align is a HIP kernel: c2m_amd.ops.roi_align.)"""
import math

import torch
import torch.nn.functional as F
from torch import nn


_adj_cache = {}      # "dense" / "csr" -> (weakref(edge_index), version, data_ptr, N, dtype, payload): eager entries only
_capture_local = {}  # the same keys for the capture in progress (tensors of that graph's private pool): dropped at its start / end


def reset_capture_cache():
    """Called by TrainStep.capture before and after a HIP-graph capture: an adjacency tensor built while capturing lives in THAT
    graph's pool and must never be handed to another capture (the second graph would read memory it does not own)."""
    _capture_local.clear()


def _cached(kind, edge_index, N, dtype, build):
    import weakref
    cap = bool(edge_index.is_cuda and torch.cuda.is_current_stream_capturing())
    store = _capture_local if cap else _adj_cache
    hit = store.get(kind)
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version and hit[2] == edge_index.data_ptr() \
            and hit[3] == N and hit[4] == dtype:
        return hit[5]
    val = build(cap)
    store[kind] = (weakref.ref(edge_index), edge_index._version, edge_index.data_ptr(), N, dtype, val)
    return val


def _edge_multiplicity(edge_index, dst, src, N, dtype):
    """A[i, j] = number of edges j -> i.  Every GATv2 layer of a step sees the same edge_index (10 calls per generator step), and
    the accumulate-index_put_ that builds A is a chain of ~8 small kernels (bounds asserts, index arithmetic, sort): built once
    per edge_index tensor (object, version and address) in eager mode and once per CAPTURE inside a HIP-graph capture (the
    tensor then lives in that graph's pool).  The accumulation adds exact 1.0s, so the result does not depend on the order."""
    return _cached("dense", edge_index, N, dtype, lambda cap: torch.zeros(N, N, dtype=dtype, device=edge_index.device).index_put_(
        (dst, src), torch.ones_like(dst, dtype=dtype), accumulate=True))


def _csr_layout(dst, N, width):
    """(order, slot) of the padded rows for a KNOWN width, device ops only (no host read-back: capturable).  The in-degrees are an
    exact integer count (one-hot sum, no atomics); rank within a row = position in the stable order minus the row's start."""
    E = dst.numel()
    order = torch.argsort(dst, stable=True)
    counts = (dst.unsqueeze(1) == torch.arange(N, device=dst.device).unsqueeze(0)).sum(0)
    start = torch.cumsum(counts, 0) - counts
    ds = dst[order]
    slot = ds * width + (torch.arange(E, device=dst.device) - start[ds])
    return order, slot, counts


def _padded_csr(edge_index, dst, src, N):
    """Edges grouped by target node into fixed-width rows: (order, slot, width) with edge order[k] sitting in row dst, column
    rank-within-row -- slot = dst * width + rank, every slot distinct.  The width (largest in-degree) is the one host-side value:
    read back once per edge_index in eager mode.  Inside a HIP-graph capture ONLY that integer is taken from the eager warm-up
    step's entry; argsort / counts / slots are recomputed by captured kernels from the static edge_index, so a replay follows new
    edges copied into it (the eager entry's tensors used to be baked into the graph -- stale after a copy, freed by the
    next eager step).  A replay whose largest in-degree exceeds the captured width would write slots of the next row: that is
    flagged by a captured comparison and raised after the replay (TrainStep._check_deferred_nan), and the slots are clamped so the
    replay itself stays in bounds."""
    def build(cap):
        if cap:
            hit = _adj_cache.get("csr")
            if hit is None or hit[3] != N:
                raise RuntimeError("GATv2Conv (> 64 nodes): run one eager step with this graph size before capturing a HIP graph "
                                   "(the padded edge layout needs the largest in-degree on the host)")
            width = hit[5][2]
            order, slot, counts = _csr_layout(dst, N, width)
            from .utils import utils as U
            U.defer_check((counts.max() > width) if dst.numel() else torch.zeros((), dtype=torch.bool, device=dst.device),
                          f"GATv2Conv: an in-degree above {width}, the row width this HIP graph was captured with "
                          "(edge_index changed: capture again)")
            return order, slot.clamp(max=N * width - 1), width
        E = dst.numel()
        counts = torch.bincount(dst, minlength=N)
        width = max(int(counts.max()) if E else 0, 1)
        order, slot, _ = _csr_layout(dst, N, width)
        return order, slot, width
    return _cached("csr", edge_index, N, torch.int64, build)


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

    DENSE_MAX_NODES = 64

    def forward(self, x, edge_index):
        N, H, C = x.shape[0], self.heads, self.out_channels
        xl = self.lin_l(x).view(N, H, C)
        xr = self.lin_r(x).view(N, H, C)
        src, dst = edge_index[0].long(), edge_index[1].long()
        if N <= self.DENSE_MAX_NODES:
            # Dense form for the few-object scene graphs of the path (3 objects x batch): logits for ALL ordered pairs, the
            # edge MULTIPLICITY matrix A[i, j] (edges j -> i) as the mask and weight of the softmax.  No scatter / index_add
            # (whose float atomics are the one run-to-run nondeterminism torch would add to the step), and a formulation
            # independent of the edge-list restatement the oracle uses (oracle/thirdparty.py::gatv2_conv), so comparing the two
            # is a real check of both.
            A = _edge_multiplicity(edge_index, dst, src, N, x.dtype)
            from . import ops
            if ops.gatv2_dense_ok(xl, N, C):
                # the chain below as one launch per direction (csrc/gnn.hip): ~18 forward and ~35 backward launches per layer on
                # [N, N, H, C] temporaries otherwise -- five layers per step, most of the object branch's launches
                return ops.gatv2_dense(xl, xr, self.att.view(H, C), A, self.negative_slope) + self.bias
            e = F.leaky_relu(xl.unsqueeze(0) + xr.unsqueeze(1), self.negative_slope)          # [i, j, H, C]
            logit = (e * self.att).sum(-1)                                                     # [i, j, H]
            logit = logit.masked_fill(A.unsqueeze(-1) == 0, float("-inf"))
            mx = logit.max(dim=1).values
            mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)                        # nodes without incoming edges
            ex = (logit - mx.unsqueeze(1).detach()).exp() * A.unsqueeze(-1)
            alpha = ex / (ex.sum(1, keepdim=True) + 1e-16)
            return torch.einsum("ijh,jhc->ihc", alpha, xl).mean(dim=1) + self.bias
        # Many nodes (> 64: B >= 22 clips of three objects): edges grouped by target node into rows of `width` slots
        # (_padded_csr), softmax and weighted sum as dense reductions along the row -- no scatter / index_add over edges, so no
        # float atomics in the forward, and the gathers' backward is torch's sort-based accumulate (deterministic).  Empty
        # slots carry -inf logits / zero messages.  tests/test_host_cpu.py holds this form to the dense one.
        order, slot, width = _padded_csr(edge_index, dst, src, N)
        so, do = src[order], dst[order]
        e = F.leaky_relu(xl[so] + xr[do], self.negative_slope)                                 # [E, H, C] in row order
        logit = (e * self.att).sum(-1)                                                         # [E, H]
        pad = torch.full((N * width, H), float("-inf"), dtype=x.dtype, device=x.device).index_copy(0, slot, logit)
        pad = pad.view(N, width, H)
        mx = pad.max(dim=1).values
        mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)                            # nodes without incoming edges
        ex = (pad - mx.unsqueeze(1).detach()).exp()                                            # exp(-inf) = 0 in empty slots
        alpha = ex / (ex.sum(1, keepdim=True) + 1e-16)                                         # [N, width, H]
        msg = torch.zeros((N * width, H, C), dtype=x.dtype, device=x.device).index_copy(0, slot, xl[so]).view(N, width, H, C)
        return (alpha.unsqueeze(-1) * msg).sum(1).mean(dim=1) + self.bias
