"""Shared glue for the block mirrors: torch.nn modules are kept ONLY as parameter containers (identical
state_dict keys / default initialisation to the reference); all arithmetic goes through c2m_amd.ops (HIP)."""
import torch

from ... import ops


def pad_triple(padding):
    """ReflectionPad3d-style padding (int or [w,w,h,h,t,t]) -> (pt, ph, pw); asymmetric pads are not used."""
    if isinstance(padding, int):
        return (padding, padding, padding)
    p = list(padding)
    if len(p) == 6:
        if p[0] != p[1] or p[2] != p[3] or p[4] != p[5]:
            raise NotImplementedError("asymmetric 3-D padding")
        return (p[4], p[2], p[0])
    if len(p) == 3:
        return tuple(p)
    raise ValueError(f"padding {padding}")


def conv_module(x, conv, act=None, padding=None, padding_mode=None, dgrad_channels=None):
    """Run an nn.Conv2d / nn.Conv3d container through the implicit-GEMM kernel."""
    pad = conv.padding if padding is None else padding
    mode = conv.padding_mode if padding_mode is None else padding_mode
    return ops.conv(x, conv.weight, conv.bias, stride=tuple(conv.stride), padding=tuple(pad) if not isinstance(pad, int) else pad,
                    padding_mode=mode, act=act, dgrad_channels=dgrad_channels)


_pending_counters = {}      # id(num_batches_tracked) -> [tensor, increments since the last flush]
_defer_counters = False     # set by GeneratorFullModel.forward for the duration of one forward


class deferred_batch_counters:
    """with deferred_batch_counters(): ... -- BatchNorm step counters of every block called inside are bumped by one
    multi-tensor add at exit (GeneratorFullModel.forward / inference); blocks used on their own bump immediately."""

    def __enter__(self):
        global _defer_counters
        self.prev, _defer_counters = _defer_counters, True

    def __exit__(self, *exc):
        global _defer_counters
        _defer_counters = self.prev
        if not self.prev:
            flush_batch_counters()


def flush_batch_counters():
    """nn.BatchNorm's `num_batches_tracked += 1` is one tiny kernel per layer call (79 per step, 0.35 ms of launches): the
    increments are collected and applied by ONE multi-tensor add at the end of the model's forward (the buffer values seen by
    state_dict() / checkpoints are the reference's; nothing reads the counter mid-forward -- momentum is a constant here)."""
    if not _pending_counters:
        return
    tensors = [v[0] for v in _pending_counters.values()]
    counts = [v[1] for v in _pending_counters.values()]
    _pending_counters.clear()
    by_dev = {}
    for t, c in zip(tensors, counts):
        by_dev.setdefault((t.device, t.dtype), ([], []))
        by_dev[(t.device, t.dtype)][0].append(t)
        by_dev[(t.device, t.dtype)][1].append(c)
    for ts, cs in by_dev.values():
        torch._foreach_add_(ts, cs)


def feeds_conv(conv, padding=None, padding_mode=None):
    """`feeds=` of the norm helpers below: the ONE convolution container that reads the norm's result (ops.conv_consumer)."""
    pad = conv.padding if padding is None else padding
    mode = conv.padding_mode if padding_mode is None else padding_mode
    return ops.conv_consumer(conv.weight, tuple(conv.stride), tuple(pad) if not isinstance(pad, int) else pad, mode, conv.bias)


def batch_norm_module(x, bn, act=None, feeds=None, private_input=False):
    """nn.BatchNorm{1,2,3}d container, train mode: batch statistics + running-stat update, fused activation.
    feeds: `feeds_conv(...)` when the result goes into exactly one convolution and nowhere else (ops.conv_consumer)."""
    if bn.training:
        if not _defer_counters:
            bn.num_batches_tracked += 1
        else:
            ent = _pending_counters.get(id(bn.num_batches_tracked))
            if ent is None:
                _pending_counters[id(bn.num_batches_tracked)] = [bn.num_batches_tracked, 1]
            else:
                ent[1] += 1
        return ops.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, act, bn.eps, bn.momentum, feeds,
                                  private_input)
    # eval: running statistics (inference path) -- scale/shift folded into the same apply kernel
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    return ops.norm_apply_eval(x, bn.running_mean, invstd, bn.weight, bn.bias, act)


def instance_norm_module(x, inorm, act=None, feeds=None, private_input=False):
    return ops.instance_norm_act(x, inorm.weight, inorm.bias, act, inorm.eps, feeds, private_input)


def fold_time(x):
    """[B,C,T,H,W] -> [T*B,C,H,W], frame-major (the reference's cat(unbind(x, 2), 0))."""
    b, c, t, h, w = x.shape
    return x.permute(2, 0, 1, 3, 4).reshape(t * b, c, h, w)


def unfold_time(x, t):
    """[T*B,C,H,W] -> [B,C,T,H,W] (the reference's cat(x.unsqueeze(2).chunk(t, 0), 2))."""
    tb, c, h, w = x.shape
    return x.reshape(t, tb // t, c, h, w).permute(1, 2, 0, 3, 4)
