"""resize_flow / resize_video / isnan / seeding helpers (reference: src/utils/utils.py:346-412)."""
import random

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import ops as _ops


def _fold(v):
    b, c, t, h, w = v.shape
    return v.permute(2, 0, 1, 3, 4).reshape(t * b, c, h, w)


def _unfold(x, t):
    tb, c, h, w = x.shape
    return x.reshape(t, tb // t, c, h, w).permute(1, 2, 0, 3, 4)


def resize_flow(flow, new_shape):
    """bilinear(align_corners=True) resize of a pixel-unit flow, magnitudes rescaled (utils.py:346-354)."""
    _, _, h, w = flow.shape
    new_h, new_w = int(new_shape[0]), int(new_shape[1])
    out = _ops.resize_bilinear(flow.detach(), (new_h, new_w), align_corners=True)
    out[:, 0] /= w / float(new_w)
    out[:, 1] /= h / float(new_h)
    return out


def resize_video(video, scale_factor, mode="nearest", is_flow=False):
    """utils.py:357-372.  At scale_factor 1 (the only value the configs use) every mode is the identity, so the
    reference's fold -> interpolate -> unfold round trip is skipped; other sizes go through the bilinear kernel."""
    if video is None:
        return None
    if not isinstance(scale_factor, (list, tuple)) and scale_factor == 1:
        return video
    t = video.shape[2]
    flat = _fold(video)
    if is_flow:
        h, w = video.shape[-2:]
        return _unfold(resize_flow(flat, [int(h * scale_factor), int(w * scale_factor)]), t)
    size = list(scale_factor) if isinstance(scale_factor, (list, tuple)) else \
        [int(video.shape[-2] * scale_factor), int(video.shape[-1] * scale_factor)]
    if mode == "bilinear" and not flat.requires_grad:
        return _unfold(_ops.resize_bilinear(flat, size, align_corners=False), t)
    return _unfold(F.interpolate(flat, size=size, mode=mode), t)


_deferred_nan_checks = []          # [(flag tensor computed INSIDE the capture, description)] of the capture in progress / last capture


def begin_deferred_nan():
    """Start collecting the NaN checks of one graph capture (TrainStep.capture); drops the previous capture's list, which
    would otherwise pin that graph's pool tensors."""
    del _deferred_nan_checks[:]


def end_deferred_nan():
    """The checks recorded since begin_deferred_nan(); the caller (TrainStep) owns and evaluates them after each replay."""
    out = list(_deferred_nan_checks)
    del _deferred_nan_checks[:]
    return out


def isnan(x, input_tensor=None):
    """Raises ValueError on NaN like the reference (utils.py:375-379) -- one host sync per call.  While the current stream is
    being captured into a HIP graph a host sync is illegal: the flag `any(isnan(x))` is computed by a captured kernel into a
    static tensor and evaluated after the replay (TrainStep._check_deferred_nan, or check_deferred_nan())."""
    if x.is_cuda and torch.cuda.is_current_stream_capturing():
        _deferred_nan_checks.append((torch.isnan(x).any(), "theta loss" if input_tensor is None else str(type(input_tensor))))
        return x
    if torch.any(torch.isnan(x)):
        raise ValueError(f"Value is nan {x}, input tensor is {input_tensor}")
    return x


def defer_check(flag, what):
    """Record a captured boolean (True = failed) of the capture in progress next to the NaN flags: evaluated after every replay."""
    _deferred_nan_checks.append((flag, what))


def check_deferred_nan():
    """Evaluate NaN checks recorded during a capture that was not made through TrainStep.capture (which keeps its own)."""
    for flag, what in _deferred_nan_checks:
        if bool(flag):
            raise ValueError(f"Value is nan ({what})")


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dist_all_reduce_tensor(tensor, reduce="mean"):
    """In-place all-reduce of a logging scalar/tensor over the ranks (utils.py:176-189); identity with one rank."""
    world = get_world_size()
    if world < 2:
        return tensor
    if reduce not in ("mean", "sum"):
        raise NotImplementedError
    with torch.no_grad():
        dist.all_reduce(tensor)
        if reduce == "mean":
            tensor /= world
    return tensor


def dist_all_gather_tensor(tensor):
    """Concatenation over ranks along dim 0 (utils.py:192-202); identity with one rank."""
    world = get_world_size()
    if world < 2:
        return tensor
    parts = [torch.empty_like(tensor) for _ in range(world)]
    with torch.no_grad():
        dist.all_gather(parts, tensor.contiguous())
    return torch.cat(parts, dim=0)


def init_cudnn(deterministic, benchmark):
    """utils.py:412-423.  The convolutions of this package are hand-written HIP kernels with a fixed summation order
    (no MIOpen, no autotuning), so the two switches only affect the few torch ops left on the path; they are forwarded
    to torch.backends.cudnn (MIOpen on ROCm) for parity of behaviour."""
    torch.backends.cudnn.deterministic = deterministic
    torch.backends.cudnn.benchmark = benchmark


def is_master():
    return get_rank() == 0


def set_random_seed(seed, by_rank=False):
    if by_rank:
        seed += get_rank()
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
