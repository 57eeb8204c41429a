"""Host side of the HIP kernels: torch.autograd Functions over the C ABI (include/c2m_hip.h).

PyTorch is used for device memory, streams and autograd bookkeeping only; every op below launches hand-written
gfx950 kernels from libc2m_hip.so on torch's current HIP stream.  There is no CPU or eager fallback: a tensor that is
not on a HIP device raises (the oracle in oracle/ is test infrastructure and is never imported from here).
"""
import collections
import ctypes
import os
import weakref
import math

import numpy as np
import torch

from . import _lib

G, WG = _lib.GEOM, _lib.WINO_GEOM      # names of the geom[] entries (include/c2m_geom.h: one definition for kernels and host)

ACT = {None: 0, "none": 0, "relu": 1, "lrelu": 2, "sigmoid": 3}
LRELU_SLOPE = 0.2


def _stream():
    """torch's current HIP stream of the current device as a raw handle (torch.cuda.current_stream() costs ~9 us of
    Python per call -- 500 calls per step; the private raw getter is the same lookup without the Stream object)."""
    return ctypes.c_void_p(_raw_stream(_cur_device()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None or _cur_device is None:          # older / newer torch without the private getters
    _raw_stream = lambda idx: torch.cuda.current_stream(idx).cuda_stream
    _cur_device = torch.cuda.current_device


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev(*ts):
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("c2m_amd ops need tensors on a HIP device (no CPU fallback by design)")
        if t.dtype != torch.float32 and t.dtype != torch.bfloat16:
            raise RuntimeError(f"c2m_amd ops take fp32 (or, on the bf16 data path, bf16) tensors, got {t.dtype}")
        if cur is None:
            cur = _cur_device()
        if t.device.index != cur:       # kernels are launched on the CURRENT device's stream (see _stream)
            raise RuntimeError(f"c2m_amd ops: tensor on cuda:{t.device.index} but the current device is cuda:{cur}; "
                               "call torch.cuda.set_device() (or use `with torch.cuda.device(...)`) first")


def _f(t):
    return t if t.is_contiguous() else t.contiguous()


BF16 = torch.bfloat16


def _dt(t):
    """Element-type code of the C ABI (csrc/dtype.h): 0 = fp32, 1 = bf16."""
    return 1 if t.dtype == torch.bfloat16 else 0


def _as(t, dtype):
    """t in `dtype`, contiguous (None passes through)."""
    if t is None:
        return None
    return _f(t if t.dtype == dtype else t.to(dtype))


def _thin(M, splits, npix, two_target=False, ncls=1):
    """The dispatch rule of c2m_conv_igemm for the <= 4-output-row vector-ALU kernels (conv_igemm.hip): those are fp32 in,
    fp32 out in either precision mode, so the host must hand them fp32 tensors."""
    return M <= 4 and splits == 1 and npix >= 16384 and not two_target and ncls == 1


# =============================================================================================== convolution
_geom_cache = {}
_conv_bf16 = False
# minimum region fill of a Winograd data-gradient launch: the padded 18x34 domain of the 16x32 reflect layers fills 53 % of its
# 8x16 regions and still beats the gather kernel by 14-30 % (tools/run_conv.py A/B); the 10x18 domain (35 %) does not
_WINO_DFIT = float(os.environ.get("C2M_WINO_DFIT", "0.5"))
_WINO_MIN_WGS = int(os.environ.get("C2M_WINO_MIN_WGS", "128"))   # smallest Winograd forward / dgrad grid: 160 workgroups of a 1536-deep layer still beat the gather kernel 1.5x
_WGRAD_WIDE_S2 = os.environ.get("C2M_WGRAD_WIDE_S2", "1") != "0"      # bf16 stride-2 weight gradient on the 16-byte-load kernel (A/B knob)
_WINO_TPAIRS = os.environ.get("C2M_WINO_TPAIRS", "1") != "0"    # 3x3x3 reflect data gradient over unpadded frames (A/B knob)
# reflect-padded 3x3 data gradients as interior (exact domain, Winograd) + pad ring (conv_ring.hip): "auto" (maps of >= C2M_RING_MIN_PIX
# pixels) | "off" (padded domain + two-target epilogue + fold, rounds 1-4) | "force" (tests: every eligible layer with H, W >= 4)
_RING = os.environ.get("C2M_RING", "auto")
_RING_BUFFER = os.environ.get("C2M_RING_BUFFER", "1") != "0"     # ring terms through a compact buffer added by the Winograd epilogue (0: in place)
_RING_MIN_PIX = int(os.environ.get("C2M_RING_MIN_PIX", "1024"))
_RING_F2_MAX_PIX = int(os.environ.get("C2M_RING_F2_MAX_PIX", "4096"))      # largest map on which an F(2x2) interior + ring beats the padded domain
_WINO4 = os.environ.get("C2M_WINO4", "auto")       # F(4x4,3x3) for the 2-D Winograd layers: "off" | "auto" (rule: _wino4_pays) | "force" (tests: every eligible 2-D Winograd launch)
_WINO = os.environ.get("C2M_WINOGRAD", "auto")      # "auto" | "off" | "force" (tests: every eligible shape)
# Winograd WEIGHT gradient (round 2: fragments built in registers from raw LDS patches): "auto" = the layers where it beats
# the direct kernel, "off", "force" (tests: every eligible shape).
_WINO_WGRAD = os.environ.get("C2M_WINOGRAD_WGRAD", "auto")      # "auto" | "off" | "force"


def set_conv_precision(precision):
    """"fp32" (default: exact fp32 MFMA, fp32 tensors) or "bf16" (BASELINE configs[2-4]): the bf16 data path.  Convolutions
    multiply bf16 operands on the bf16 MFMA with fp32 accumulation AND write bf16 activations (NCHW, as PyTorch lays them
    out); norm / activation / up-sampling / pooling / warping kernels read and write bf16 with fp32 arithmetic in registers;
    gradients of those activations are bf16 as well.  Stays fp32: weights and their gradients (the optimizer's master copy),
    norm statistics, losses, flows / occlusion / index masks (bit-exact paths), the <= 4-channel heads (flow, occlusion,
    RGB).  An fp32 tensor handed to a bf16-mode convolution is cast once (RNE), so old call sites keep working.
    Returns the previous setting."""
    global _conv_bf16
    if precision not in ("fp32", "bf16"):
        raise ValueError(f"unknown conv precision {precision!r}")
    prev = "bf16" if _conv_bf16 else "fp32"
    _conv_bf16 = precision == "bf16"
    return prev


class conv_precision:
    """Context manager form of set_conv_precision (the backward of a conv uses the precision of its forward)."""

    def __init__(self, precision):
        self.precision = precision

    def __enter__(self):
        self.prev = set_conv_precision(self.precision)

    def __exit__(self, *exc):
        set_conv_precision(self.prev)


class ConvProfiler:
    """Optional HIP-event timing of every implicit-GEMM launch (bench.py roofline): events are recorded on the stream
    the kernels are launched on (torch's current stream).  Off by default; zero overhead when off."""
    active = None

    def __init__(self):
        self.records = []          # (kind, flops, start_event, end_event, tag, algorithmic bytes)
        self._pool, self._used = [], 0

    def __enter__(self):
        ConvProfiler.active = self
        return self

    def __exit__(self, *exc):
        ConvProfiler.active = None

    def __del__(self):
        try:
            L = _lib.lib()
            for e in self._pool:
                L.c2m_event_destroy(e)
        except Exception:
            pass

    def _event(self):
        """HIP timing event without the system-scope fence of torch.cuda.Event (csrc/events.hip)."""
        if self._used == len(self._pool):
            import ctypes
            h = ctypes.c_void_p(None)
            _lib.check(_lib.lib().c2m_event_create(ctypes.addressof(h)), "event_create")
            self._pool.append(h.value)
        self._used += 1
        return self._pool[self._used - 1]

    @staticmethod
    def _ms(e0, e1):
        import ctypes
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.lib().c2m_event_elapsed_ms(e0, e1, ctypes.addressof(ms)), "event_elapsed")
        return float(ms.value)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, flops, e0, e1, _, nbytes in self.records:
            d = out.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += self._ms(e0, e1)
        return out

    def table(self):
        """Per launch-shape totals: [(tag, launches, ms, TFLOP/s)] sorted by time."""
        torch.cuda.synchronize()
        agg = {}
        for kind, flops, e0, e1, tag, _ in self.records:
            d = agg.setdefault((kind,) + tuple(tag), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += self._ms(e0, e1)
            d[2] += flops
        rows = [(k, v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12) for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r[2])


# weight gradient on a side stream: "auto" = bf16 data path AND inside a HIP-graph capture; "1" = bf16 and fp32, eager too; "0" off.
# Measured: graph replays of the bf16 configurations 89.7 -> 86.7 ms (configs[2]), 159.4 -> 152.0 (configs[4]),
# 55.5 -> 55.2 (configs[3]).  Eager: fork / join cost the host ~25 us per conv node (events, stream switch, record_stream), which a
# host-bound eager step pays in full (configs[3] 62.0 -> 65.3 ms; configs[4], GPU-bound, 161.6 -> 155.1), and per-kernel HIP-event
# times of overlapping launches are no kernel measurements any more -- so eager steps stay on one stream.  fp32 configs[1]:
# 74.5 -> 75.3 ms (the fp32 Winograd kernels fill the CUs' LDS and registers by themselves).
_WGRAD_SIDE = os.environ.get("C2M_WGRAD_STREAM", "auto")
_side_streams = {}


def _side_stream(device):
    # (stream priorities measured, round 5: side stream high 62.9 vs 62.1 ms, low = default; the step on a prioritised non-default
    # stream 62.1 (high) / 63.9 (normal) vs 62.2 on the default stream -- nothing to gain, default priorities everywhere)
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return s


# ---- auxiliary stream for the object branch (round 5).  The RoI head of the appearance encoder and the object GNN are ~600 launches
# of a few microseconds on [24, 1024]-sized tensors per step (Linear layers, GATv2 attention, theta losses): ~3 ms of GPU time that
# uses a sliver of the chip and, with gt thetas (use_gt_training: True), feeds nothing but the theta losses.  Run on a second stream
# they execute NEXT TO the convolutions of the motion encoders / decoder / generator instead of between them -- forward, and
# backward too: autograd runs every backward node on the stream of its forward and orders the streams itself.  No arithmetic
# changes; every kernel pairing is one the concurrency stress test covers.  C2M_AUX_STREAM=0 switches it off (A/B).
_AUX = os.environ.get("C2M_AUX_STREAM", "1")          # 1 | 0 | obj (= roi + gnn) | novgg | roi | gnn | vgg (parts: diagnosis, A/B)
_aux_streams = {}        # (device index, lane) -> stream.  Lane 0: the object branch; lane 1: the ground-truth VGG pass (losses.py)
_aux_open = {}           # (device index, lane) -> the lane has work the main stream has not joined yet


class aux_branch:
    """`with ops.aux_branch(*input_tensors): ...` -- the body's launches go to an auxiliary stream of the device, ordered behind
    everything the current stream has been given so far.  The results may only be used on the main stream after `aux_join`."""

    def __init__(self, *inputs, part="", lane=0):
        self.inputs = [t for t in inputs if torch.is_tensor(t) and t.is_cuda]
        self.ctx = None
        self.lane = lane
        self.on = _AUX == "1" or _AUX == part or (_AUX == "obj" and part in ("roi", "gnn")) or \
            (_AUX.startswith("no") and _AUX[2:] != part)
        # bench.py's per-launch HIP events time ISOLATED launches: the steps that carry them run on one stream (a kernel that shares
        # the chip with another stream's is stretched, in the events and in a rocprofv3 trace alike)
        self.on = self.on and ConvProfiler.active is None

    def __enter__(self):
        if not self.on or not self.inputs:
            if self.inputs:
                aux_join(*self.inputs, lanes=(self.lane,))    # (diagnosis mode, an earlier part on this lane: this one reads its results)
            return self
        dev = self.inputs[0].device
        key = (dev.index, self.lane)
        aux = _aux_streams.get(key)
        if aux is None:
            aux = _aux_streams[key] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        if aux == main:
            return self
        aux.wait_stream(main)                     # (a second block of the same branch: aux order kept, the new dependencies added)
        for t in self.inputs:
            t.record_stream(aux)                  # allocated on the main stream's pool, read by aux-stream kernels
        _aux_open[key] = True
        self.ctx = torch.cuda.stream(aux)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# ---- deferred weight gradients (round 5): see _ConvFn.backward
_DEFER_WGRAD = os.environ.get("C2M_DEFER_WGRAD", "1") != "0"
_defer = {"on": False, "devs": set(), "seen": set(), "hold": collections.deque(), "grads": {}, "by_w": {}}


class deferred_wgrads:
    """`with ops.deferred_wgrads(): loss.backward()` -- weight gradients of the convolutions are computed on the side stream and
    joined once, when the block exits (on the stream that is current then).  Parameter gradients must not be read inside the block."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and _DEFER_WGRAD

    def __enter__(self):
        _defer["on"] = self.enabled
        _defer["seen"].clear()
        _defer["grads"].clear()
        _defer["by_w"].clear()
        return self

    def __exit__(self, *exc):
        _defer["on"] = False
        for idx in _defer["devs"]:
            side = _side_streams[idx]
            torch.cuda.current_stream(side.device).wait_stream(side)
        _defer["devs"].clear()
        _defer["hold"].clear()            # (the joined stream orders every later write behind the side stream's reads)
        _defer["grads"].clear()
        return False


def deferred_grad_stream(grad):
    """The side stream if `grad` is a weight / bias gradient that a deferred launch of the current backward is still producing
    (ops.deferred_wgrads), else None.  For whoever touches a parameter gradient INSIDE backward (the gradient reducer's hook)."""
    return _defer["grads"].get(grad.data_ptr()) if _defer["grads"] else None


def _record_stream_all(stream, *objs):
    """Every device tensor in objs (tensors, NC8 placeholders and their forms, dicts / tuples of them) is in use on `stream`."""
    for o in objs:
        if o is None:
            continue
        if torch.is_tensor(o):
            if o.is_cuda:
                o.record_stream(stream)
                form = getattr(o, "_c2m_nc8", None)               # (version, NC8 form) riding on the tensor object
                if form is not None and torch.is_tensor(form[1]):
                    form[1].record_stream(stream)
                ent = _nc8_only.get(o.data_ptr()) if getattr(o, "_c2m_nc8_only", False) else None
                if ent is not None and torch.is_tensor(ent[2]):
                    ent[2].record_stream(stream)
        elif isinstance(o, dict):
            _record_stream_all(stream, *o.values())
        elif isinstance(o, (tuple, list)):
            _record_stream_all(stream, *o)


def branch_streams(device):
    """The streams -- besides the one backward() is called on -- that backward nodes of this package run on (the object branch's
    auxiliary stream, the weight-gradient side stream).  Whoever consumes gradients from ANOTHER stream before backward() has
    returned (the gradient reducer's communication stream) has to wait for these as well."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return [s for (i, _), s in _aux_streams.items() if i == idx] + ([_side_streams[idx]] if idx in _side_streams else [])


def _on_aux_stream(dev):
    cur = None
    for (i, _), s in _aux_streams.items():
        if i == dev.index:
            cur = torch.cuda.current_stream(dev) if cur is None else cur
            if s == cur:
                return True
    return False


def aux_join(*outputs, lanes=None):
    """The current stream waits for the auxiliary streams' work (all lanes, or the given ones); `outputs` (tensors made there) may
    be used from here on."""
    for key, is_open in list(_aux_open.items()):
        if not is_open or (lanes is not None and key[1] not in lanes):
            continue
        aux = _aux_streams[key]
        main = torch.cuda.current_stream(aux.device)
        if aux != main:
            main.wait_stream(aux)
            _record_stream_all(main, *outputs)
        _aux_open[key] = False


def _timed(kind, flops, fn, tag=(), nbytes=0):
    prof = ConvProfiler.active
    if prof is None:
        return fn()
    L, st = _lib.lib(), _stream()
    e0, e1 = prof._event(), prof._event()
    _lib.check(L.c2m_event_record(e0, st), "event_record")
    rc = fn()
    _lib.check(L.c2m_event_record(e1, st), "event_record")
    prof.records.append((kind, flops, e0, e1, tag, nbytes))
    return rc


def _ceil(a, b):
    return (a + b - 1) // b * b


def _cdiv(a, b):
    return (a + b - 1) // b


def _triple(v, nd):
    if isinstance(v, int):
        return (1,) * (3 - nd) + (v,) * nd if nd < 3 else (v, v, v)
    v = tuple(int(a) for a in v)
    return (1,) * (3 - len(v)) + v if len(v) < 3 else v


def _pad3(v, nd):
    if isinstance(v, int):
        return (0,) * (3 - nd) + (v,) * nd
    v = tuple(int(a) for a in v)
    return (0,) * (3 - len(v)) + v


def _choose_ck(C, taps):
    """Channels per tap inside one 16-deep K-step (CK x NS = 16): the padding-minimal choice, larger CK on ties."""
    best = None
    for ck in (16, 8, 4):
        ns = 16 // ck
        cost = _ceil(C, ck) * _ceil(taps, ns) * (1.0 + 0.03 * (ns - 1))
        if best is None or cost < best[0] * 0.97:
            best = (cost, ck)
    return best[1]


def _patch_ok(C, taps3, stride, splits, Ho, Wo, M):
    """LDS-patch kernel eligibility: 3x3 (kt = 1) stride-1, whole 16-channel chunks without much padding, a grid big
    enough not to need split-K, output rows/cols that fill the (rows x 32) tile reasonably."""
    kt, kh, kw = taps3
    if (kt, kh, kw) != (1, 3, 3) or tuple(stride) != (1, 1, 1) or M <= 4:      # <= 4 rows: vector-ALU kernels
        return False
    best = _ceil(C, _choose_ck(C, 9))
    if _ceil(C, 16) > 1.10 * best:
        return False
    rows = 8 if M <= 32 else 4
    return Wo >= 32 and Ho >= rows and (_ceil(Wo, 32) * _ceil(Ho, rows)) <= 1.15 * Wo * Ho


def _patch_bf16_ok(C, stride, Ho, Wo, M, fill_limit=None, nc8=True):
    """bf16 LDS-patch kernel (8 rows x 32 columns per workgroup): 3x3 stride-1.  Where it beats the bf16 gather kernel
    (per-shape A/B of BASELINE configs[2], gpurun_out/r02/table_cfg2*.txt): deep reductions (>= 128 channels: the weight
    image of a chunk is re-streamed per 256 pixels, which only pays with many chunks) or few output rows (<= 64), and
    tiles that do not hang far over the (padded) domain; <= 4 output rows stay on the vector-ALU kernels."""
    if tuple(stride) != (1, 1, 1) or M <= 4 or C < 12 or Wo < 32 or Ho < 8 or _ceil(C, 16) > 1.25 * C:
        return False
    fill = (_ceil(Wo, 32) * _ceil(Ho, 8)) / float(Wo * Ho)
    if _NC8 and nc8:      # the channel-blocked kernel (conv_nc8.hip) has no 2-byte gathers to amortise: every layer whose tiles fit
                          # (nc8: the plan can run it -- planes of whole 8-pixel groups; otherwise the NCHW kernel's own rule below)
        return fill <= (fill_limit or _NC8_FILL)      # (the padded 34x66 domain of a reflect data gradient fills 58 % of its tiles: still 1.5x the gather kernel)
    if C < 128:
        return (M <= 64 or _PATCH_SMALLC) and fill <= 1.2
    return fill <= _PATCH_FILL[0] or (fill <= _PATCH_FILL[1] and C >= 256)


_NC8_VARIANT = int(os.environ.get("C2M_NC8_VARIANT", "0"))      # tile / buffering variant of conv_patch_nc8_kernel (0 = the library's rule)
# norm kernels write the NC8 side output for the conv behind / in front.  OFF by default: measured on one box (configs[3] / [2] graph
# replays) it removes 77 of 174 layout passes (-0.95 ms) but the 8-channel x 8-pixel apply kernels run 2x as long as the per-plane
# ones they replace (+1.2 ms): 47.7 vs 47.05 ms, 71.5 vs 70.7 ms.  Kept as an A/B knob with its tests (tests/test_gpu_nc8.py).
_NC8_NORM = os.environ.get("C2M_NC8_NORM", "0") != "0"
# NC8 gather kernel (conv_gather_nc8_kernel, conv_igemm.hip): every bf16 forward / data gradient the NC8 patch forms do not take
_NC8_LOG = None
_NC8_LOG_TAG = [""]          # (tools/nc8_producers.py: what made the gradient tensor a backward conversion sees)
_NC8_GRAD = os.environ.get("C2M_NC8_GRAD", "1") != "0"      # act_bwd / tap backward write NC8 only where every reader is an NC8 kernel
_NC8_POISON = os.environ.get("C2M_NC8_POISON", "0") == "1"  # debug: the unused NCHW storage of such gradients is NaN
_G8 = os.environ.get("C2M_G8", "1") != "0"
_G8_VARIANT = int(os.environ.get("C2M_G8_VARIANT", "0"))
_NC8_3D = os.environ.get("C2M_NC8_3D", "1") != "0"           # bf16 3x3x3 layers on the NC8 kernels (A/B knob)
_NC8_S2_WGRAD_MIN_PIX = 16384
_NC8_S2 = os.environ.get("C2M_NC8_S2", "1") != "0"           # bf16 4x4 stride-2 forward on the parity-plane kernel (A/B knob)
# padded domains of the reflect data gradients: tile waste above which the NC8 gather form takes them (kernel times on one box,
# patch kernel with the buffer-store two-target epilogue vs gather form: 66x130 domains, fill 1.34: 78 vs 111 us and 144 vs 151;
# 34x66, fill 1.71: 106 vs 87; 18x34, fill 2.5: 68 vs 45)
_NC8_DGRAD_FILL = float(os.environ.get("C2M_NC8_DGRAD_FILL", "1.5"))
_NC8_S2_FILL_G8 = float(os.environ.get("C2M_NC8_S2_FILL_G8", "1.5"))      # stride-2 patch forms: largest tile waste where the gather form is the alternative
_NC8_FILL = float(os.environ.get("C2M_NC8_FILL", "2.6"))      # (18x34 padded domains of the 16x32 reflect data gradients: 2.5; still 1.5x+ the gather kernel)
_NC8_WGRAD = os.environ.get("C2M_NC8_WGRAD", "1") != "0"      # bf16 3x3 weight gradient from NC8 operands (A/B knob)
_NC8 = os.environ.get("C2M_NC8", "1") != "0"        # bf16 3x3 stride-1 layers on channel-blocked input (A/B knob)
_PATCH_FILL = tuple(float(v) for v in os.environ.get("C2M_PATCH_FILL", "1.15,1.4").split(","))     # tuning knobs (A/B runs)
_PATCH_SMALLC = os.environ.get("C2M_PATCH_SMALLC", "0") == "1"


def _pack_bf16_patch(w, M, C, s_m, s_c, mode=0):
    """c2m_pack_weights_bf16_patch: contiguous native 3x3 weights -> bf16 [chunk][tap][Mpad][16]; mode 2: native 4x4 weights of a
    stride-2 layer -> [chunk][input parity][2x2 tap][Mpad][16] (conv_nc8.hip, S2)."""
    L = _lib.lib()
    nbytes = L.c2m_pack_weights_bf16_s2_bytes(M, C) if mode >= 2 else L.c2m_pack_weights_bf16_patch_bytes(M, C)
    out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    g = np.array([M, C, s_m, s_c, mode], dtype=np.int64)
    _lib.check(L.c2m_pack_weights_bf16_patch(_p(w), _p(out), _gp(g), _stream()), "pack_weights_bf16_patch")
    out._c2m_job = (1, g)                   # how to rebuild this pack in place (refresh_trainable_packs)
    return out


_BF16_SPLIT_DIV = float(os.environ.get("C2M_BF16_SPLIT_DIV", "4"))      # bf16 kernels: a quarter of the fp32 rule's K splits (A/B on configs[2,3], graph mode: 1 -> 60.0 / 93.7 ms, 2 -> 58.4 / 92.5, 4 -> 58.0 / 91.2, no splits -> 65.4 / 98.2)


def _splits(L, M, nk, npix, bf16):
    """c2m_conv_igemm_splits, optionally thinned out for the bf16 kernels (every split still owns >= 1 K-step)."""
    S = L.c2m_conv_igemm_splits(M, nk, npix)
    if bf16 and _BF16_SPLIT_DIV != 1 and S > 1:
        S = max(1, int(S / _BF16_SPLIT_DIV))
        S = _cdiv(nk, _cdiv(nk, S))
    return S


def _patch_splits(L, M, C, npix, bf16=False):
    """Split-K count for the patch kernel: whole 16-channel chunks per split, every split non-empty."""
    nch = _cdiv(C, 16)
    s0 = _splits(L, M, nch * 9, npix, bf16)
    return _cdiv(nch, _cdiv(nch, s0))


_nc8_only = {}      # storage address -> (weakref(NC8-only tensor), its _version, NC8 form)


def _drop_nc8_only(key):
    ent = _nc8_only.get(key)
    if ent is not None and ent[0]() is None:
        del _nc8_only[key]


def _mark_nc8_only(t, tn):
    """`t` is an NCHW-shaped tensor object whose storage is never written: its values exist as `tn` (NC8) only."""
    t._c2m_nc8 = (t._version, tn)
    t._c2m_nc8_only = True
    key = t.data_ptr()
    _nc8_only[key] = (weakref.ref(t), t._version, tn)
    weakref.finalize(t, _drop_nc8_only, key)
    return t


def _to_nc8(x, keep=None):
    """c2m_nchw_to_nc8: contiguous bf16 [N, C, H, W] -> [N, ceil(C/8), H, W, 8] (the conv input form of conv_nc8.hip).
    The result is remembered ON the source tensor (attribute, with the tensor's `_version`: any in-place change invalidates it),
    so an activation that feeds several convolutions -- the SPADE maps, the shared conditioning features -- is converted once;
    keep: a dict that additionally pins the result per source tensor for the duration of one backward node (the data gradient and
    the weight gradient share the NC8 form of dY even when autograd hands them the gradient as a fresh tensor object)."""
    hit = getattr(x, "_c2m_nc8", None)
    if hit is not None and hit[0] == x._version and hit[1].device == x.device:
        if keep is not None:
            keep[x.data_ptr()] = (x, hit[1])       # (a tensor that exists in NC8 form only must never depend on the attribute alone)
        return hit[1]
    if keep is not None:
        hit = keep.get(x.data_ptr())
        if hit is not None and hit[0] is x:
            return hit[1]
    # NC8-only tensors: the NCHW storage of such a tensor was never written.  If the attribute did not travel with the
    # tensor object (tensor hooks, retain_grad, saved_tensors_hooks hand out another object over the same storage) the NC8 form is
    # recovered from the registry by storage address + version; a tensor tagged NC8-only without any valid form is an error, never
    # a layout pass over uninitialised memory.
    ent = _nc8_only.get(x.data_ptr())
    if ent is not None:
        src = ent[0]()
        if src is not None and src.shape == x.shape and src.dtype == x.dtype and ent[1] == x._version and ent[2].device == x.device:
            if keep is not None:
                keep[x.data_ptr()] = (x, ent[2])
            return ent[2]
    if getattr(x, "_c2m_nc8_only", False):
        raise RuntimeError("c2m_amd: a tensor that exists in NC8 form only reached a convolution without a valid NC8 form (it was "
                           "modified in place, or its single-consumer contract -- feeds= / private_input= -- was broken)")
    N, C = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])                 # (H, W), or (T, H, W): the pixel axis of the layout pass is everything behind C
    if _NC8_LOG is not None:                # tools/nc8_producers.py: who produced the tensors that need a layout pass
        fn = x.grad_fn
        _NC8_LOG.append((type(fn).__name__ if fn is not None else ("leaf" if torch.is_grad_enabled() else "backward" + _NC8_LOG_TAG[0]), tuple(x.shape)))
    y = torch.empty((N, _cdiv(C, 8)) + sp + (8,), device=x.device, dtype=BF16)
    _lib.check(_lib.lib().c2m_nchw_to_nc8(_p(x), _p(y), N, C, int(np.prod(sp)), _stream()), "nchw_to_nc8")
    if keep is not None:
        keep[x.data_ptr()] = (x, y)
    x._c2m_nc8 = (x._version, y)
    return y


def _time_pair_table_kt(T, reflect):
    """Data gradient of a 3-tap pad-1 convolution in time: output frame t sums dY frame `to` through time tap kt for every
    (to, kt) with reflect(to + kt - 1) == t (reflect padding: the pad frames' contributions land on the frames they mirror) or
    to + kt - 1 == t (zeros).  int32 [T][11] = {npairs, (to, kt) x 5} -- c2m_conv3d_dgrad_nc8's ptab."""
    tab = np.zeros((T, 11), dtype=np.int32)
    for t in range(T):
        pairs = []
        for kt in range(3):
            for to in range(T):
                q = to + kt - 1
                if reflect:
                    q = -q if q < 0 else (2 * T - 2 - q if q >= T else q)
                if q == t:
                    pairs.append((to, kt))
        assert 1 <= len(pairs) <= 5
        tab[t, 0] = len(pairs)
        for j, (to, kt) in enumerate(pairs):
            tab[t, 1 + 2 * j], tab[t, 2 + 2 * j] = to, kt
    return tab


def _pack_bf16_k333(w, M, C, dgrad=False, cin_total=None):
    """Weights [Cout][Cin][3][3][3] of a 3x3x3 layer -> three c2m_pack_weights_bf16_patch images back to back (time tap kt = 0, 1, 2);
    rows M = output channels, reduction C = input channels -- or, for the data gradient, rows = input channels over output channels."""
    L = _lib.lib()
    nb = L.c2m_pack_weights_bf16_patch_bytes(M, C)
    out = torch.empty(3 * nb, device=w.device, dtype=torch.uint8)
    wf = w.reshape(-1)
    for kt in range(3):
        g = np.array([M, C, 27, (cin_total or M) * 27, 0] if dgrad else [M, C, C * 27, 27, 0], dtype=np.int64)
        _lib.check(L.c2m_pack_weights_bf16_patch(_p(wf[kt * 9:]), _p(out[kt * nb:]), _gp(g), _stream()), "pack_weights_bf16_patch (3-D)")
    return out


def _nc8_launch(L, A, x, dst, y2, b, geom, act, slope, keep=None):
    """One 3x3 stride-1 launch on channel-blocked input: the layout pass + c2m_conv_patch_nc8 (timed together)."""
    xn = _to_nc8(x, keep)
    geom[G.NC8_VARIANT] = _NC8_VARIANT
    return L.c2m_conv_patch_nc8(_p(A), _p(xn), _p(dst), _p(y2), _p(b), _gp(geom), act, slope, _stream())


def _pack_bf16_gather(w, M, C, kdims, stride, s_m, s_c):
    """c2m_pack_weights_bf16_gather: contiguous native weights -> prod(stride) class images [tap][16-channel chunk][half][Mpad] of
    16-byte bf16 units (the NC8 gather kernel's K order: tap-major)."""
    L = _lib.lib()
    kt, kh, kw = kdims
    st, sh, sw = stride
    g = np.array([M, C, 16, kt, kh, kw, st, sh, sw, s_m, s_c], dtype=np.int64)
    nbytes = L.c2m_pack_weights_bf16_gather_bytes(_gp(g))
    if nbytes <= 0:
        raise RuntimeError("c2m_pack_weights_bf16_gather_bytes: bad geometry")
    out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    _lib.check(L.c2m_pack_weights_bf16_gather(_p(w), _p(out), _gp(g), _stream()), "pack_weights_bf16_gather")
    out._c2m_job = (2, g)
    return out


def _g8_bm(M):
    """Row tile of conv_gather_nc8_kernel for M output rows (the rule of c2m_conv_igemm's NC8 gather form)."""
    return 32 if M <= 32 else (64 if (M <= 64 or 1 <= M % 128 <= 64) else 128)


_G8_SPLIT_WGS = int(os.environ.get("C2M_G8_SPLIT_WGS", "320"))


def _g8_splits(M, nk, npix, ncls=1):
    """K splits of an NC8 gather launch (256-pixel tiles): only launches that leave CUs empty are split, towards ~1.25 workgroups
    per CU with >= 8 K-steps per split -- at bf16 MFMA rates the fp32 slabs (4 bytes per output and split, written and read back)
    cost more than a second round of workgroups saves."""
    tiles = _cdiv(M, _g8_bm(M)) * _cdiv(npix, 256) * ncls
    if tiles >= 200 or nk < 16:
        return 1
    S = min(_cdiv(_G8_SPLIT_WGS, tiles), nk // 8, 16)
    if S < 2:
        return 1
    return _cdiv(nk, _cdiv(nk, S))


def _g8_geom(geom, C, taps, S):
    """geom of the same launch on the NC8 gather form: nk = taps * ceil(C/16) K-steps in (tap, chunk) order."""
    g = geom.copy()
    nk = taps * _cdiv(C, 16)
    g[[G.NK, G.LDA, G.NS, G.SPLITS, G.CIN, G.TAPS, G.PRECISION, G.PATCH, G.G8, G.G8_VARIANT]] = (nk, nk * 16, 1, S, C, taps, 1, 0, 1, _G8_VARIANT)
    return g


def _g8_taps(offs, device):
    t = np.zeros((len(offs), 4), dtype=np.int32)
    t[:, :3] = np.asarray(offs, dtype=np.int32).reshape(len(offs), 3)
    t[:, 3] = 1
    return torch.from_numpy(t.reshape(-1)).to(device)


def _set_patch(geom, iy0, ix0, pty, ptx):
    geom[G.PATCH] = 1
    geom[G.PATCH_IY0], geom[G.PATCH_IX0] = iy0, ix0
    geom[G.PATCH_TY:G.PATCH_TY + 3] = pty
    geom[G.PATCH_TX:G.PATCH_TX + 3] = ptx


def _tap_offsets(kt, kh, kw, off_t, off_y, off_x):
    """(dt, dy, dx) per tap in (t, y, x) row-major order."""
    return [(int(off_t[a]), int(off_y[b]), int(off_x[c])) for a in range(kt) for b in range(kh) for c in range(kw)]


def _kstep_table(C, tap_offs, chan_stride, ck, extra_groups=0, ones_group=False):
    """K-step / row-group table: [(groups), 1 + NS, 4] int32 in (channel chunk, tap group) order.
    group = header {chan_off, nvalid, 0, 0} + NS x {dt, dy, dx, valid}; optional ones group (nvalid = -2) and zero pads."""
    ns = 16 // ck
    taps = len(tap_offs)
    nch, ntg = _cdiv(C, ck), _cdiv(taps, ns)
    n = nch * ntg
    tab = np.zeros((n + (1 if ones_group else 0) + extra_groups, 1 + ns, 4), dtype=np.int32)
    offs = np.zeros((ntg * ns, 4), dtype=np.int32)
    offs[:taps, :3] = np.asarray(tap_offs, dtype=np.int32).reshape(taps, 3)
    offs[:taps, 3] = 1
    for ch in range(nch):
        tab[ch * ntg:(ch + 1) * ntg, 0, 0] = ch * ck * chan_stride
        tab[ch * ntg:(ch + 1) * ntg, 0, 1] = min(ck, C - ch * ck)
        tab[ch * ntg:(ch + 1) * ntg, 1:, :] = offs.reshape(ntg, ns, 4)
    if ones_group:
        tab[n, 0, 1] = -2
    return tab, nch, ntg


def _pack_rows(wm, ck):
    """[M, C, taps] -> contiguous [M, nk*16] in the kernel's K order (chunk, tap group, tap slot, channel)."""
    M, C, taps = wm.shape
    ns = 16 // ck
    nch, ntg = _cdiv(C, ck), _cdiv(taps, ns)
    if nch * ck != C or ntg * ns != taps:
        buf = wm.new_zeros(M, nch * ck, ntg * ns)
        buf[:, :C, :taps] = wm
        wm = buf
    # .contiguous(): in degenerate cases (one tap, one chunk) the reshape chain is a strided VIEW of the weights
    return wm.reshape(M, nch, ck, ntg, ns).permute(0, 1, 3, 4, 2).reshape(M, nch * ntg * 16).contiguous()


def _geom(**kw):
    g = np.zeros(G.LEN, dtype=np.int64)      # CLS_PO .. LEN - 1: per-class (po_t, po_y, po_x) of a class-batched launch
    idx = dict(M=G.M, nk=G.NK, lda=G.LDA, Npix=G.NPIX, To=G.TO, Ho=G.HO, Wo=G.WO, Ti=G.TI, Hi=G.HI, Wi=G.WI, st=G.ST, sh=G.SH, sw=G.SW,
               in_sn=G.IN_SN, in_st=G.IN_ST, in_sh=G.IN_SH, out_sn=G.OUT_SN, out_sc=G.OUT_SC, out_st=G.OUT_ST, out_sh=G.OUT_SH,
               out_sw=G.OUT_SW, out_off=G.OUT_OFF, reflect=G.REFLECT, is3d=G.IS3D, ns=G.NS, in_sc=G.IN_SC, splits=G.SPLITS,
               slab_stride=G.SLAB_STRIDE, Cin=G.CIN, taps=G.TAPS, ntg=G.NTG, ngroups=G.NGROUPS, x_bytes=G.X_BYTES, dy_bytes=G.DY_BYTES)
    for k, v in kw.items():
        g[idx[k]] = v
    return g


def _wino4_pays(L, M, K, nimg, Ho, Wo):
    """F(4x4,3x3) (conv_wino4.hip: 64 rows x 16x32 outputs per 512-thread workgroup, ONE workgroup per CU) instead of F(2x2,3x3)
    for a 2-D Winograd launch with M output rows, K input channels over nimg Ho x Wo output domains.  Measured per shape on one box
    (tools/bench_wino4.py, round 3): 1.25-1.42x on grids of >= 2.5 workgroups per CU that fill their 16x32 regions, 1.08-1.12x on
    the deep 16x32 layers, 0.87-0.92x on the 320-workgroup launches of 128-row layers at 32x64."""
    if _WINO4 == "force":
        return True
    if _WINO4 != "auto":
        return False
    regions = nimg * L.c2m_wino4_regions(Ho, Wo)
    fill = nimg * Ho * Wo / float(regions * 512)
    # (the deep 16x32 layers -- 512 -> 512 / 256 -> 256, 320 / 160 workgroups -- measure 1.08x / 1.12x alone but made the step
    # slower when routed here: 67.7 -> 68.3 ms; they stay on F(2x2))
    # fill: full regions only.  The padded 66 x 130 domains of the reflect data gradients (0.67 of their 16x32 regions) measure
    # 1.06-1.24x as isolated forward grids (tools/bench_wino4.py) but routing them here made the step slower (67.8 -> 68.1 ms)
    wgs = regions * (M // 64) if M % 64 == 0 else 0
    if K < _WINO4_MIN_K or M < 64 or fill < _WINO4_MIN_FILL:
        return False
    return wgs >= _WINO4_MIN_WGS or (K >= 256 and M >= 256 and wgs >= _WINO4_DEEP_WGS)


_WINO4_MIN_K = int(os.environ.get("C2M_WINO4_MIN_K", "64"))
_WINO4_DEEP_WGS = int(os.environ.get("C2M_WINO4_DEEP_WGS", "1000000"))      # deep (K, M >= 256) layers on small maps: A/B knob, off by default
_WINO4_MIN_FILL = float(os.environ.get("C2M_WINO4_MIN_FILL", "0.9"))
_WINO4_MIN_WGS = int(os.environ.get("C2M_WINO4_MIN_WGS", "600"))      # 640 workgroups (256 -> 256 at 32x64, 40 images): 1.12x; 320: 0.8-0.95x


def _wino4_filter(w, Cout, Cin, dgrad):
    """c2m_wino4_filter_transform: native [Cout][Cin][3][3] -> packed 6x6 U = G g G^T fragments of conv_wino4_kernel."""
    L = _lib.lib()
    M, K = (Cin, Cout) if dgrad else (Cout, Cin)
    U = torch.empty(L.c2m_wino4_upack_floats(M, K), device=w.device, dtype=torch.float32)
    _lib.check(L.c2m_wino4_filter_transform(_p(w), _p(U), Cout, Cin, dgrad, _stream()), "wino4_filter_transform")
    return U


def _ring_pack(w, Cout, Cin):
    """c2m_ring_pack: native [Cout][Cin][3][3] -> the A fragments of reflect_ring_dgrad_kernel (four sides x three taps)."""
    L = _lib.lib()
    A = torch.empty(L.c2m_ring_pack_floats(Cout, Cin), device=w.device, dtype=torch.float32)
    _lib.check(L.c2m_ring_pack(_p(w), _p(A), Cout, Cin, _stream()), "ring_pack")
    return A


def _wino_geom(head, To=0, in_st=0, out_st=0, cin=0, nkt=0, toff=0, Ti=0, treflect=0):
    """geom[] of c2m_conv_wino (include/c2m_hip.h): 33 entries; the tail describes the time taps of a 3x3x3 layer."""
    g = np.zeros(WG.LEN, dtype=np.int64)
    assert len(head) in (WG.X_BYTES + 1, WG.EXT_X + 1)       # M .. X_BYTES in enum order (+ the Y_interior block Y2_SN .. EXT_X)
    g[:len(head)] = head
    g[WG.TO:WG.TREFLECT + 1] = (To, in_st, out_st, cin, nkt, toff, Ti, treflect)
    return g


def _time_pair_table(T):
    """Temporal reflect padding (pad 1, 3 taps) folded into the data gradient: output frame t of the unpadded axis sums dY
    frame `to` through time tap kt for every (to, kt) with reflect(to + kt - 1) == t.  int32 [T][11] = {npairs, (to, U block =
    2 - kt: the flipped tap order of the packed data-gradient weights) x 5} -- c2m_conv_wino geom[33]."""
    tab = np.zeros((T, 11), dtype=np.int32)
    for t in range(T):
        pairs = []
        for kt in range(3):
            for to in range(T):
                q = to + kt - 1
                q = -q if q < 0 else (2 * T - 2 - q if q >= T else q)
                if q == t:
                    pairs.append((to, 2 - kt))
        assert 2 <= len(pairs) <= 5
        tab[t, 0] = len(pairs)
        for j, (to, ub) in enumerate(pairs):
            tab[t, 1 + 2 * j], tab[t, 2 + 2 * j] = to, ub
    return tab


class _ConvPlan:
    """Everything shape-dependent for one conv layer geometry (cached): gather tables + geom arrays."""

    def __init__(self, xs, ws, stride, pad, reflect, device, bf16=False, dgrad_rows=None):
        L = _lib.lib()
        self.bf16 = bf16
        nd = len(xs) - 2
        self.is3d = is3d = 1 if nd == 3 else 0
        N, Cin = xs[0], xs[1]
        if ws[1] != Cin:
            # torch.nn.functional.conv2d raises here too.  Without the check the pack kernels read Cin * taps weights per output
            # row out of a tensor that holds ws[1] * taps: past its end -- harmless next to other cached blocks, a GPU page fault
            # when the weight sits at the end of an allocator segment (the intermittent abort of round 5's suite: a test that
            # passed a 32-channel input to a 16-channel weight)
            raise RuntimeError(f"convolution: input of shape {tuple(xs)} has {Cin} channels, weight of shape {tuple(ws)} expects "
                               f"{ws[1]} (groups are not on the C2M path)")
        Ti, Hi, Wi = (xs[2], xs[3], xs[4]) if nd == 3 else (1, xs[2], xs[3])
        Cout = ws[0]
        kt, kh, kw = (ws[2], ws[3], ws[4]) if nd == 3 else (1, ws[2], ws[3])
        st, sh, sw = stride
        pt, ph, pw = pad
        To, Ho, Wo = (Ti + 2 * pt - kt) // st + 1, (Hi + 2 * ph - kh) // sh + 1, (Wi + 2 * pw - kw) // sw + 1
        if min(To, Ho, Wo) <= 0:
            raise ValueError("empty convolution output")
        if kt * kh * kw > 128:
            raise NotImplementedError(f"conv kernels with more than 128 taps ({kt}x{kh}x{kw}) are not on the C2M path "
                                      "(c2m_pack_weights / the gather tables hold 128 tap offsets)")
        if reflect and (pt >= Ti and pt > 0 or ph >= Hi and ph > 0 or pw >= Wi and pw > 0):
            raise ValueError("reflect padding must be smaller than the input extent")
        if 4 * N * max(Cin * Ti * Hi * Wi, Cout * To * Ho * Wo) >= 2 ** 31:
            raise ValueError("tensor too large: the gather uses 32-bit byte offsets (< 2 GiB per tensor)")
        self.dims = (N, Cin, Cout, Ti, Hi, Wi, To, Ho, Wo, kt, kh, kw)
        # dgrad_rows: only the first dgrad_rows input channels need a gradient (the rest of a concatenated input is a
        # tensor without grad, e.g. the rastered sparse motion in front of final_fuse): the data-gradient GEMM then has
        # dM rows instead of Cin -- 32 instead of 34 there, which is one 32-row MFMA tile instead of a half-empty 64
        dM = self.dM = int(dgrad_rows) if dgrad_rows and tuple(stride) == (1, 1, 1) and 0 < dgrad_rows < Cin else Cin
        self.stride, self.pad, self.reflect = stride, pad, reflect
        self.out_shape = (N, Cout, To, Ho, Wo) if nd == 3 else (N, Cout, Ho, Wo)
        taps = kt * kh * kw
        self.K = Cin * taps
        in_sc, osp = Ti * Hi * Wi, To * Ho * Wo
        # ---- Winograd F(2x2,3x3) for the 3x3 stride-1 2-D layers (fp32 mode): forward, and the data gradient when the
        # padding is zeros (the reflect data gradient runs over the padded domain with the two-target epilogue)
        self.wino_fwd = self.wino_dgrad = self.wino_wgrad = self.ring_dgrad = False
        self.wino4_fwd = self.wino4_dgrad = False         # F(4x4,3x3) instead of F(2x2,3x3) for that launch (2-D layers)
        if not bf16 and (kt, kh, kw) == (1, 3, 3) and tuple(stride) == (1, 1, 1) and (ph, pw) == (1, 1) and nd == 2:
            wrows = 32 if Cout <= 32 else 64                      # workgroup tile: 64 (32 for Cout <= 32) output x 32 input channels
            wg_tiles = _cdiv(Cin, 32) * _cdiv(Cout, wrows)
            # A/B against the direct kernel (tools/ab_wino_wgrad.py, AB_EXTRA=1 for the marginal shapes): the Winograd form
            # wins 1.1-1.3x even on a half-empty 64-row tile as long as there are two tiles
            waste = 32.0 * wrows * wg_tiles / (Cin * Cout)
            if Hi % 2 == 0 and Wi % 16 == 0 and (Cin * Cout) % 4 == 0 and (_WINO_WGRAD == "force" or (
                    _WINO_WGRAD == "auto" and (waste <= 1.35 or (waste <= 2.0 and wg_tiles >= 2))
                    and N * (Hi // 2) * (Wi // 16) >= 8 * max(1, (768 if wrows == 32 else 512) // wg_tiles))):
                self.wino_wgrad = True
                self.wino_wg_splits = L.c2m_wino_wgrad_splits(Cout, Cin, N, Hi, Wi)
            # regions per image: 8 x 16 outputs, or the th x tw tile shape c2m_conv_wino picks for a badly fitting domain
            regions = N * L.c2m_wino_regions(Ho, Wo)
            fit = N * Ho * Wo >= 0.8 * regions * 128
            # rows: 64-row tiles need >= 48 output channels to pay; 17..32 run on the 32-row variant (MT = 1, three
            # workgroups per CU), which beats the direct kernels' 32-row tiles
            rows_ok = lambda m: m >= 48 or 17 <= m <= 32
            if _WINO == "force" or (_WINO == "auto" and fit and Cin >= 32 and rows_ok(Cout) and
                                    regions * _cdiv(Cout, 64) >= _WINO_MIN_WGS):
                self.wino_fwd = True
                self.wino_fwd_geom = _wino_geom([Cout, Cin, N, Hi, Wi, Ho, Wo, -1, -1, int(reflect), Cin * in_sc, in_sc, Wi,
                                                 Cout * osp, osp, Wo, 0, 4 * N * Cin * in_sc])
                self.wino4_fwd = _wino4_pays(L, Cout, Cin, N, Ho, Wo)
            # data gradient: zero padding -> the unpadded domain; reflect padding -> EITHER the exact H x W domain (the interior
            # of the padded gradient = the zero-padded "same" data gradient, full Winograd regions) + the pad ring folded into dX
            # by c2m_reflect_ring_dgrad (conv_ring.hip: four thin GEMMs, 3/9 (2H+2W)/(HW) of the layer's FLOPs -- maps of
            # >= _RING_MIN_PIX pixels), OR the padded (H+2)x(W+2) domain with the two-target epilogue (interior straight into dX,
            # pad ring into a scratch tensor that is then folded)
            self.ring_dgrad = False
            if reflect and dM == Cin and _RING != "off" and Hi >= 4 and Wi >= 4 and (_RING == "force" or Hi * Wi >= _RING_MIN_PIX):
                eregions = N * L.c2m_wino_regions(Hi, Wi)
                efit = N * Hi * Wi >= 0.8 * eregions * 128
                w4e = _wino4_pays(L, Cin, Cout, N, Hi, Wi)
                # measured per shape in the step (profiles/r05_ab_ring_*): the route pays where the exact domain moves the layer to
                # F(4x4) (1.3-1.65x) and on the <= 32 x 64 maps whose padded domain wastes a third of the F(2x2) regions (1.14-1.26x);
                # F(2x2) layers on >= 64 x 128 maps (padded fill 0.8-0.9) gain less than the ring launch costs
                pays = _RING == "force" or w4e or Hi * Wi <= _RING_F2_MAX_PIX
                if pays and (_WINO == "force" or (_WINO == "auto" and efit and Cout >= 32 and rows_ok(Cin) and
                                                  eregions * _cdiv(Cin, 64) >= _WINO_MIN_WGS)):
                    self.wino_dgrad = self.ring_dgrad = True
                    self.wino_dgrad_geom = _wino_geom(
                        [Cin, Cout, N, Ho, Wo, Hi, Wi, -1, -1, 0, Cout * osp, osp, Wo, Cin * in_sc, in_sc, Wi, 0, 4 * N * Cout * osp])
                    self.wino4_dgrad = w4e
            Hd, Wd = (Hi + 2, Wi + 2) if reflect else (Hi, Wi)
            dregions = N * L.c2m_wino_regions(Hd, Wd)
            dfit = N * Hd * Wd >= _WINO_DFIT * dregions * 128
            if not self.ring_dgrad and dM == Cin and (_WINO == "force" or (_WINO == "auto" and dfit and Cout >= 32 and rows_ok(Cin) and
                                                                         dregions * _cdiv(Cin, 64) >= _WINO_MIN_WGS)):
                self.wino_dgrad = True
                o = -2 if reflect else -1
                self.wino_dgrad_geom = _wino_geom(
                    [Cin, Cout, N, Ho, Wo, Hd, Wd, o, o, 0, Cout * osp, osp, Wo, Cin * Hd * Wd, Hd * Wd, Wd, 0,
                     4 * N * Cout * osp, Cin * in_sc, in_sc, Wi, 1, 1, Hi, Wi])
                self.wino4_dgrad = _wino4_pays(L, Cin, Cout, N, Hd, Wd)
        # ---- 3x3x3 stride-1 pad-1 layers (fuse_convs, the 3-D blocks): a 2-D Winograd over virtual input channels
        # (time tap, channel) with image = (sample, frame) -- the K loop is three times as deep as the 2-D layer's, the
        # temporal padding is a per-tap frame index (reflected, or a zero-record descriptor).  Data gradient: virtual
        # channels (flipped time tap, output channel) of dY over the padded (T+2, H+2, W+2) domain (reflect; then one
        # fold pass) or the unpadded one (zeros).
        self.wino3d = self.wino_wgrad3d = self.wino3d_pairs = False
        if not bf16 and nd == 3 and (kt, kh, kw) == (3, 3, 3) and tuple(stride) == (1, 1, 1) and (pt, ph, pw) == (1, 1, 1):
            rows_ok = lambda m: m >= 48 or 17 <= m <= 32
            regions = N * To * L.c2m_wino_regions(Ho, Wo)
            fit = N * To * Ho * Wo >= 0.8 * regions * 128
            hw_i, hw_o = Hi * Wi, Ho * Wo
            if _WINO == "force" or (_WINO == "auto" and fit and 3 * Cin >= 32 and rows_ok(Cout) and
                                    regions * _cdiv(Cout, 64) >= _WINO_MIN_WGS):
                self.wino_fwd = self.wino3d = True
                self.wino_fwd_geom = _wino_geom(
                    [Cout, 3 * Cin, N * To, Hi, Wi, Ho, Wo, -1, -1, int(reflect), Cin * in_sc, in_sc, Wi, Cout * osp, osp, Wo,
                     0, 4 * N * Cin * in_sc], To=To, in_st=hw_i, out_st=hw_o, cin=Cin, nkt=3, toff=-1, Ti=Ti,
                    treflect=int(reflect))
            Td, Hd, Wd = (Ti + 2, Hi + 2, Wi + 2) if reflect else (Ti, Hi, Wi)
            dregions = N * Td * L.c2m_wino_regions(Hd, Wd)
            dfit = N * Td * Hd * Wd >= _WINO_DFIT * dregions * 128
            if _WINO == "force" or (_WINO == "auto" and dfit and 3 * Cout >= 32 and rows_ok(dM) and
                                    dregions * _cdiv(dM, 64) >= _WINO_MIN_WGS):
                self.wino_dgrad = self.wino3d = True
                o = -2 if reflect else -1
                # reflect in time: launched over the Ti real frames with the pad frames' contributions as extra (frame, tap)
                # pairs of the frames they mirror onto (kernel: WinoP::ptab) -- not over Ti + 2 frames + a fold in time
                self.wino3d_pairs = bool(reflect and Cout % 8 == 0 and To == Ti and Ti >= 2 and _WINO_TPAIRS)
                Tl = Ti if self.wino3d_pairs else Td
                self.wino_dgrad_target = (N, Cin, Tl, Hd, Wd)
                self.wino_dgrad_geom = _wino_geom(
                    [dM, 3 * Cout, N * Tl, Ho, Wo, Hd, Wd, o, o, 0, Cout * osp, osp, Wo, Cin * Tl * Hd * Wd, Tl * Hd * Wd, Wd,
                     0, 4 * N * Cout * osp], To=Tl, in_st=hw_o, out_st=Hd * Wd, cin=Cout, nkt=3, toff=o, Ti=To, treflect=0)
                if self.wino3d_pairs:
                    self.wino_pair_tab = torch.from_numpy(_time_pair_table(Ti).reshape(-1)).to(device)
                    self.wino_dgrad_geom[WG.PTAB] = self.wino_pair_tab.data_ptr()
                    # every launched frame is a real one: the two-target epilogue of the 2-D layers applies per frame (interior
                    # straight into dX, only the pad ring into the scratch tensor; then the border-only fold)
                    self.wino_dgrad_geom[WG.Y2_SN:WG.EXT_X + 1] = (Cin * in_sc, in_sc, Wi, 1, 1, Hi, Wi)
            # weight gradient: the 2-D Winograd wgrad kernel over images (sample, frame) and virtual channels (kt, ci)
            wrows = 32 if Cout <= 32 else 64
            wg_tiles = _cdiv(3 * Cin, 32) * _cdiv(Cout, wrows)
            waste = 32.0 * wrows * wg_tiles / (3 * Cin * Cout)
            if Hi % 2 == 0 and Wi % 16 == 0 and (3 * Cin * Cout) % 4 == 0 and (_WINO_WGRAD == "force" or (
                    _WINO_WGRAD == "auto" and (waste <= 1.35 or (waste <= 2.0 and wg_tiles >= 2))
                    and N * Ti * (Hi // 2) * (Wi // 16) >= 8 * max(1, (768 if wrows == 32 else 512) // wg_tiles))):
                self.wino_wgrad = self.wino_wgrad3d = True
                self.wino_wg_splits = L.c2m_wino_wgrad_splits(Cout, 3 * Cin, N * Ti, Hi, Wi)
        # ---- forward
        self.ck = ck = _choose_ck(Cin, taps)
        self.fwd_patch = False
        # channel-blocked input (conv_nc8.hip): 2-D bf16 patch layers whose planes are whole 8-pixel groups
        self.nc8 = bool(bf16 and _NC8 and nd == 2 and (Hi * Wi) % 8 == 0 and (Ho * Wo) % 8 == 0)
        # ... the 3x3x3 stride-1 pad-1 layers as (sample, frame) images with (time tap, channel) chunks: forward and weight gradient
        # (channel padding up to 45 %: final_fuse has 34 input channels = three 16-channel chunks, still 2x the gather kernel)
        k333 = bool(bf16 and _NC8 and _NC8_3D and nd == 3 and (kt, kh, kw) == (3, 3, 3) and tuple(stride) == (1, 1, 1) and
                    (pt, ph, pw) == (1, 1, 1) and (Hi * Wi) % 8 == 0 and Ti >= 2 and Cin >= 12 and _ceil(Cin, 16) <= 1.45 * Cin and
                    Wo >= 32 and Ho >= 8 and (_ceil(Wo, 32) * _ceil(Ho, 8)) <= _NC8_FILL * Wo * Ho)
        self.k333_nc8 = k333 and Cout > 4
        self.k333_wgrad_nc8 = k333 and _NC8_WGRAD and Cout >= 64 and Cin >= 16
        pd = 1 if reflect else 0
        self.k333_dgrad_nc8 = bool(k333 and Cout >= 12 and _ceil(Cout, 16) <= 1.25 * Cout and dM > 4 and
                                   (_ceil(Wi + 2 * pd, 32) * _ceil(Hi + 2 * pd, 8)) <= _NC8_FILL * (Wi + 2 * pd) * (Hi + 2 * pd))
        if self.k333_dgrad_nc8:
            self.k333_ptab = torch.from_numpy(_time_pair_table_kt(Ti, bool(reflect)).reshape(-1)).to(device)
        # ... the 4x4 stride-2 pad-1 layers on the parity-plane form of the patch kernel (forward).  Half-filled tiles (8 x 16 output maps)
        # go to the NC8 gather form where it can take the layer: 40 vs 59 us on 256 -> 512 at 16x32, but 42 vs 36 us on 128 -> 256 at
        # 32x64 whose tiles are full (profiles/r04_ab_nc8.txt)
        # (only launches with >= 4096 output pixels: below that the gather form needs deep K splits -- configs[2]'s 20-image batches of
        # the same layers measured +0.7 ms with the gather form, configs[3]'s 40-image batches -0.7 ms)
        g8_in = bool(_G8 and Cin >= 12 and _ceil(Cin, 16) <= 1.45 * Cin and in_sc % 8 == 0 and N * osp >= 4096)
        g8_out = bool(_G8 and Cout >= 12 and _ceil(Cout, 16) <= 1.45 * Cout and osp % 8 == 0 and N * osp >= 4096)
        self.s2_nc8 = bool(self.nc8 and _NC8_S2 and (kt, kh, kw) == (1, 4, 4) and tuple(stride) == (1, 2, 2) and (ph, pw) == (1, 1)
                           and Hi % 2 == 0 and Wi % 2 == 0 and Cout > 4 and Cin >= 12 and Wo >= 16 and Ho >= 4 and
                           (_ceil(Wo, 32) * _ceil(Ho, 8)) <= (_NC8_S2_FILL_G8 if g8_in else _NC8_FILL) * Wo * Ho)
        # ... their data gradient (four output parity classes over one shared dY patch, one launch)
        qh, qw = Hi // 2 + (1 if reflect else 0), Wi // 2 + (1 if reflect else 0)
        self.s2_dgrad_nc8 = bool(self.nc8 and _NC8_S2 and (kt, kh, kw) == (1, 4, 4) and tuple(stride) == (1, 2, 2) and
                                 (ph, pw) == (1, 1) and Hi % 2 == 0 and Wi % 2 == 0 and Cin >= 8 and Cout >= 12 and qw >= 16 and
                                 dM == Cin and (_ceil(qw, 32) * _ceil(qh, 8)) <= (_NC8_S2_FILL_G8 if (g8_out and dM > 4) else _NC8_FILL) * qw * qh)
        self.s2_wgrad_nc8 = bool(self.nc8 and _NC8_S2 and _NC8_WGRAD and (kt, kh, kw) == (1, 4, 4) and tuple(stride) == (1, 2, 2) and
                                 (ph, pw) == (1, 1) and Hi % 2 == 0 and Wi % 2 == 0 and Cout >= 64 and Cin >= 16 and
                                 N * Ho * Wo >= _NC8_S2_WGRAD_MIN_PIX)      # (the few-pixel encoder tails: one or two chunks per split, slabs dominate)
        # ... and the weight gradient of the 3x3 stride-1 pad-1 layers from the NC8 forms of X and dY (transposed LDS reads)
        self.wgrad_nc8 = bool(self.nc8 and _NC8_WGRAD and (kt, kh, kw) == (1, 3, 3) and tuple(stride) == (1, 1, 1) and
                              (ph, pw) == (1, 1) and Cout >= 64 and Cin >= 16)      # (Cout = 32: half of a 64-row tile is padding -- 88 vs 145 TF/s on the NCHW kernel)
        if (kt, kh, kw) == (1, 3, 3) and (_patch_bf16_ok(Cin, stride, Ho, Wo, Cout, None, self.nc8) if bf16 else
                                          _patch_ok(Cin, (kt, kh, kw), stride, 1, Ho, Wo, Cout)):
            self.fwd_patch, self.ck = True, 16
            ck = 16
        ns = 16 // ck
        offs = _tap_offsets(kt, kh, kw, np.arange(kt) - pt, np.arange(kh) - ph, np.arange(kw) - pw)
        tab, nch, ntg = _kstep_table(Cin, offs, in_sc, ck)
        nk = nch * ntg
        self.nk = nk
        self.fwd_tab = torch.from_numpy(tab.reshape(-1)).to(device)
        self.fwd_splits = _patch_splits(L, Cout, Cin, N * osp, bf16) if self.fwd_patch else \
            _splits(L, Cout, nk, N * osp, bf16)
        self.fwd_geom = _geom(M=Cout, nk=nk, lda=nk * 16, Npix=N * osp, To=To, Ho=Ho, Wo=Wo, Ti=Ti, Hi=Hi, Wi=Wi, st=st,
                              sh=sh, sw=sw, in_sn=Cin * in_sc, in_st=Hi * Wi, in_sh=Wi, out_sn=Cout * osp, out_sc=osp,
                              out_st=Ho * Wo, out_sh=Wo, out_sw=1, out_off=0, reflect=int(reflect), is3d=is3d, ns=ns,
                              in_sc=in_sc, splits=self.fwd_splits, slab_stride=N * Cout * osp,
                              x_bytes=4 * N * Cin * in_sc)
        self.fwd_geom[[G.CIN, G.TAPS, G.NTG]] = (Cin, taps, ntg)
        if kt == 1 and kh == kw and (st, sh, sw) == (1, 1, 1):
            self.fwd_geom[G.SQUARE_KW] = kw                # row-major square tap set, dx ascending (thin row-blocked kernel)
        if self.fwd_patch:
            _set_patch(self.fwd_geom, -ph, -pw, (0, 1, 2), (0, 1, 2))
            if bf16:
                self.fwd_geom[G.LDA] = _ceil(Cout, 128)       # lda = padded row count of the bf16 weight image
        # ---- wgrad: same (chunk, tap group) row order + a ones group (bias gradient) + zero groups up to the tile
        self.J = L.c2m_conv_wgrad_rows(Cout, nk + 1)
        wtab, _, _ = _kstep_table(Cin, offs, in_sc, ck, extra_groups=self.J // 16 - nk - 1, ones_group=True)
        self.wg_tab = torch.from_numpy(wtab.reshape(-1)).to(device)
        self.wg_geom = self.fwd_geom.copy()
        self.wg_geom[G.PATCH] = 0
        self.wg_geom[[G.M, G.NK, G.OUT_SN, G.OUT_SC]] = (Cout, self.J, Cout * osp, osp)      # (wgrad: J rows, dY strides)
        self.wg_geom[[G.CIN, G.TAPS, G.NTG, G.NGROUPS]] = (Cin, taps, ntg, nk)
        self.wg_geom[G.DY_BYTES] = 4 * N * Cout * osp
        # bf16 data path: the 16-byte-load weight-gradient kernel (conv_wgrad_wide_bf16_kernel) takes layers whose 8-pixel
        # groups stay inside an input row up to one pad pixel: unit x stride, same-width output, |tap dx| <= 1
        self.wg_geom[G.WGRAD_WIDE] = int(bf16 and sw == 1 and Wi == Wo and Wo % 8 == 0 and Wi >= 8 and kw in (1, 3)
                               and pw == (kw - 1) // 2 and osp % 8 == 0)
        # ... and its stride-2 form (4-wide taps, pad 1: dx = -1 .. 2; every second element of a 16-element run)
        if bf16 and _WGRAD_WIDE_S2 and sw == 2 and Wi == 2 * Wo and Wo % 8 == 0 and kw == 4 and pw == 1 and osp % 8 == 0:
            self.wg_geom[G.WGRAD_WIDE] = 2
        self.wg_splits = L.c2m_conv_wgrad_splits(Cout, self.J, N * osp)
        # ---- dgrad: one launch per stride-parity class
        Tp, Hp, Wp = (Ti + 2 * pt, Hi + 2 * ph, Wi + 2 * pw) if reflect else (Ti, Hi, Wi)
        self.dgrad_target = (N, Cin, Tp, Hp, Wp)
        self.dgrad_needs_zero = False
        self.classes = []

        def dim_classes(I, O, k, s, p):
            Ip = I + 2 * p if reflect else I
            out = []
            for r in range(s):
                A = len(range(r, k, s))
                if reflect:
                    qmin, qmax, off = 0, (Ip - 1 - r) // s, r
                else:
                    qmin = max(0, -((r - p) // s))      # ceil((p - r)/s)
                    qmax = (I - 1 + p - r) // s
                    off = s * qmin + r - p
                Q = qmax - qmin + 1
                if Q <= 0:
                    continue
                if A == 0:
                    self.dgrad_needs_zero = True
                    continue
                out.append((r, A, qmin, Q, off))
            return out

        tgt_numel = N * Cin * Tp * Hp * Wp
        for (rt, At, qt, Qt, offt) in dim_classes(Ti, To, kt, st, pt):
            for (ry, Ay, qy, Qy, offy) in dim_classes(Hi, Ho, kh, sh, ph):
                for (rx, Ax, qx, Qx, offx) in dim_classes(Wi, Wo, kw, sw, pw):
                    ctaps = At * Ay * Ax
                    cck = _choose_ck(Cout, ctaps)
                    cpatch = False
                    if (At, Ay, Ax) == (1, 3, 3) and (st, sh, sw) == (1, 1, 1) and \
                            (_patch_bf16_ok(Cout, (1, 1, 1), Qy, Qx, dM, _NC8_DGRAD_FILL if (reflect and _G8) else None,
                                            not is3d and (Ho * Wo) % 8 == 0) if bf16 else
                             _patch_ok(Cout, (At, Ay, Ax), (1, 1, 1), 1, Qy, Qx, dM)):
                        cpatch, cck = True, 16
                    coffs = _tap_offsets(At, Ay, Ax, qt - np.arange(At), qy - np.arange(Ay), qx - np.arange(Ax))
                    ctab, cnch, cntg = _kstep_table(Cout, coffs, osp, cck)
                    cnk = cnch * cntg
                    npix = N * Qt * Qy * Qx
                    geom = _geom(M=dM, nk=cnk, lda=cnk * 16, Npix=npix, To=Qt, Ho=Qy, Wo=Qx, Ti=To, Hi=Ho, Wi=Wo, st=1,
                                 sh=1, sw=1, in_sn=Cout * osp, in_st=Ho * Wo, in_sh=Wo, out_sn=Cin * Tp * Hp * Wp,
                                 out_sc=Tp * Hp * Wp, out_st=st * Hp * Wp, out_sh=sh * Wp, out_sw=sw,
                                 out_off=offt * Hp * Wp + offy * Wp + offx, reflect=0, is3d=is3d, ns=16 // cck,
                                 in_sc=osp, splits=1, slab_stride=tgt_numel, x_bytes=4 * N * Cout * osp)
                    # two-target epilogue (reflect): padded coord = q*stride + r per dim; interior = [pad, pad + extent)
                    geom[G.PS_T:G.Y2_SH + 1] = (st, sh, sw, offt, offy, offx, pt, ph, pw, Ti, Hi, Wi, Cin * in_sc, in_sc, Hi * Wi, Wi)
                    geom[[G.CIN, G.TAPS, G.NTG]] = (Cout, ctaps, cntg)
                    if At == 1 and Ay == Ax and (st, sh, sw) == (1, 1, 1):
                        geom[G.SQUARE_KW] = -Ax            # data gradient: tap offsets run q - arange(A): dx descending
                    if cpatch:
                        _set_patch(geom, qy - 2, qx - 2, (2, 1, 0), (2, 1, 0))
                        if bf16:
                            geom[G.LDA] = _ceil(dM, 128)
                    self.classes.append(dict(r=(rt, ry, rx), taps=ctaps, ck=cck, nk=cnk, npix=npix, patch=cpatch,
                                             tab=torch.from_numpy(ctab.reshape(-1)).to(device), geom=geom, offs=coffs))
        # common split count for all classes (they share one slab set); fall back to 1 if they cannot agree
        S = min(_splits(L, dM, c["nk"], c["npix"], bf16) for c in self.classes) if self.classes else 1
        if S > 1:
            for c in self.classes:
                if _cdiv(c["nk"], _cdiv(c["nk"], S)) != S:
                    S = 1
                    break
        if any(c["patch"] for c in self.classes):        # stride-1 3x3: a single class
            S = _patch_splits(L, dM, Cout, self.classes[0]["npix"], bf16)
        self.dgrad_splits = S
        for c in self.classes:
            c["geom"][G.SPLITS] = S
        # Algorithmic FLOPs (roofline bookkeeping): the data gradient of a convolution has the MACs of its forward
        # (2 * Cout * K * output pixels -- FlopCounterMode's count).  Reflect-padded layers LAUNCH over the
        # padded domain; every launch is credited with its share of the forward count, not with the padded volume.
        self.fwd_flops = 2.0 * Cout * self.K * N * osp
        self.dgrad_flops = self.fwd_flops * dM / Cin
        self.dgrad_work = float(sum(c["taps"] * c["npix"] for c in self.classes)) or 1.0
        # ---- class batching: the stride parity classes of a k % s == 0 conv share every dimension (same taps per
        # class, same Q extents) and differ only in weights, tap table and output origin -> ONE launch
        # (blockIdx.z = class), which fills the chip where a single class (1/s^d of the pixels) cannot
        self.cls_batch = None
        cl = self.classes
        ncls = st * sh * sw
        self.classes_packable = len(cl) == ncls and kt % st == 0 and kh % sh == 0 and kw % sw == 0 and \
            len({(c["ck"], c["taps"]) for c in cl}) == 1 and \
            [c["r"] for c in cl] == [(a, b, c_) for a in range(st) for b in range(sh) for c_ in range(sw)]
        if 1 < ncls <= 8 and len(cl) == ncls and kt % st == 0 and kh % sh == 0 and kw % sw == 0 and \
                not self.dgrad_needs_zero and not any(c["patch"] for c in cl) and \
                len({(c["nk"], c["ck"], c["taps"]) for c in cl}) == 1:
            key = lambda c: (c["npix"],) + tuple(int(v) for v in c["geom"][G.TO:G.WO + 1])
            runs, i = [], 0
            while i < ncls:                       # maximal runs of consecutive classes with identical extents
                j = i
                while j + 1 < ncls and key(cl[j + 1]) == key(cl[i]):
                    j += 1
                runs.append((i, j + 1))
                i = j + 1
            nk0, ck0 = cl[0]["nk"], cl[0]["ck"]
            SB = min(_splits(L, Cin, nk0, cl[a]["npix"] * (b - a), bf16) for a, b in runs)
            groups = []
            for a, b in runs:
                g = cl[a]["geom"].copy()
                g[G.SPLITS] = SB
                g[G.NCLS], g[G.A_CLS], g[G.KTAB_CLS] = b - a, Cin * nk0 * 16, nk0 * (1 + 16 // ck0)
                assert b - a <= 8                     # C2M_G_MAX_CLS (checked against the flag entries in c2m_geom.h)
                for i in range(a, b):
                    g[G.CLS_OUT_OFF + i - a] = cl[i]["geom"][G.OUT_OFF]
                    po = G.CLS_PO + 3 * (i - a)
                    g[po:po + 3] = cl[i]["geom"][G.PO_T:G.PO_X + 1]
                groups.append(dict(geom=g, tab=torch.cat([c["tab"] for c in cl[a:b]]), ncls=b - a, first=a,
                                   npix=cl[a]["npix"]))
            if len(groups) < ncls:
                self.cls_batch = dict(groups=groups, ncls=ncls, ck=ck0, taps=cl[0]["taps"], nk=nk0)
                self.dgrad_splits = SB
        # ---- NC8 gather form (conv_gather_nc8_kernel) of every bf16 forward / data gradient the NC8 patch forms above do not take:
        # the same launches (geometry, classes, two-target epilogue, split-K) with channel-blocked input, a compact tap table and
        # bf16 weights in (tap, chunk) order.  Channel padding up to 45 % (34 -> 48); <= 4 output rows stay on the vector-ALU kernels.
        chan_ok = lambda c: c >= 12 and _ceil(c, 16) <= 1.45 * c
        g8 = bool(bf16 and _NC8 and _G8)
        self.g8_fwd = bool(g8 and not (self.k333_nc8 or self.s2_nc8 or (self.fwd_patch and self.nc8)) and Cout > 4 and chan_ok(Cin)
                           and in_sc % 8 == 0 and taps <= 64)
        if self.g8_fwd:
            self.g8_fwd_splits = _g8_splits(Cout, taps * _cdiv(Cin, 16), N * osp)
            self.g8_fwd_geom = _g8_geom(self.fwd_geom, Cin, taps, self.g8_fwd_splits)
            self.g8_fwd_tab = _g8_taps(offs, device)
        patch_nc8 = any(c["patch"] for c in cl) and not is3d and (Ho * Wo) % 8 == 0
        self.g8_dgrad = bool(g8 and cl and not (self.k333_dgrad_nc8 or self.s2_dgrad_nc8 or patch_nc8) and dM > 4 and chan_ok(Cout)
                             and osp % 8 == 0 and self.classes_packable and max(c["taps"] for c in cl) <= 64 and
                             not any(c["patch"] for c in cl))
        if self.g8_dgrad:
            ctaps = cl[0]["taps"]
            nk8 = ctaps * _cdiv(Cout, 16)
            self.g8_a_cls = nk8 * 2 * _ceil(dM, 128) * 16          # bytes of one class image
            if self.cls_batch is not None:
                S8 = min(_g8_splits(dM, nk8, grp["npix"], grp["ncls"]) for grp in self.cls_batch["groups"])
            else:
                S8 = min(_g8_splits(dM, nk8, c["npix"]) for c in cl)
            if S8 > 1 and _cdiv(nk8, _cdiv(nk8, S8)) != S8:
                S8 = 1
            self.g8_dgrad_splits = S8
            for c in cl:
                c["g8"] = _g8_geom(c["geom"], Cout, ctaps, S8)
                c["g8tab"] = _g8_taps(c["offs"], device)
            for grp in (self.cls_batch["groups"] if self.cls_batch else ()):
                grp["g8"] = _g8_geom(grp["geom"], Cout, ctaps, S8)
                grp["g8"][G.A_CLS], grp["g8"][G.KTAB_CLS] = 0, 0
                grp["g8tab"] = torch.cat([c["g8tab"] for c in cl[grp["first"]:grp["first"] + grp["ncls"]]])


def _bwd_reads_only_nc8(pl, need_x, need_w):
    """True when every kernel of this bf16 layer's backward reads the output gradient in its NC8 form (conv_nc8.hip / the NC8 gather
    form): the gradient then need not exist in NCHW at all (`_virtual_grad`)."""
    if not (pl.bf16 and _NC8 and _NC8_GRAD and (pl.dims[6] * pl.dims[7] * pl.dims[8]) % 8 == 0):
        return False
    if need_x:
        patch_nc8 = bool(pl.classes) and all(c["patch"] for c in pl.classes) and not pl.is3d and (pl.dims[7] * pl.dims[8]) % 8 == 0
        if not (pl.k333_dgrad_nc8 or pl.s2_dgrad_nc8 or pl.g8_dgrad or patch_nc8):
            return False
    if need_w and not (pl.k333_wgrad_nc8 or pl.wgrad_nc8 or pl.s2_wgrad_nc8):
        return False
    return True


def _fwd_reads_only_nc8(pl, need_w):
    """True when the forward AND the weight gradient of this bf16 layer read the input in its NC8 form only (the data gradient never
    reads the input): an activation whose single consumer is this layer need not exist in NCHW (`feeds=` of the norm ops)."""
    if not (pl.bf16 and _NC8 and _NC8_GRAD and (pl.dims[3] * pl.dims[4] * pl.dims[5]) % 8 == 0):
        return False
    if not (pl.k333_nc8 or pl.s2_nc8 or (pl.fwd_patch and pl.nc8) or pl.g8_fwd):
        return False
    if need_w and not (pl.k333_wgrad_nc8 or pl.wgrad_nc8 or pl.s2_wgrad_nc8):
        return False
    return True


def _consumer_reads_only_nc8(like, feeds):
    """`feeds` = conv_consumer(...) and `like` a tensor of the consumer's input shape / dtype: True when that convolution will read
    its input through NC8 kernels only AND in one piece -- `conv()` runs tensors past 2 GiB as batch chunks, i.e. on VIEWS of its
    input, which do not carry the NC8 form."""
    cw, cstride, cpad, cmode = feeds[:4]
    cb = feeds[4] if len(feeds) > 4 else None
    nd = like.dim() - 2
    stride3, pad3 = _triple(cstride, nd), _pad3(cpad, nd)
    if _chunks_for_2gib(like.shape, cw.shape, stride3, pad3) > 1:
        return False
    cpl = _plan(like, _f(cw), stride3, pad3, cmode == "reflect" and any(pad3), None)
    # (a frozen weight with a trainable bias still runs _ConvFn._wgrad -- the bias gradient comes out of the weight-gradient kernel)
    return _fwd_reads_only_nc8(cpl, cw.requires_grad or (cb is not None and cb.requires_grad))


def conv_consumer(conv_w, stride=1, padding=0, padding_mode="zeros", conv_b=None):
    """`feeds=` argument of batch_norm_act / instance_norm_act / spade_norm_act: the ONE convolution that consumes the op's result
    (the caller guarantees nothing else reads it -- `out = norm(x); out = conv(out)` inside a block).  Where that layer's forward and
    weight gradient run on NC8 kernels the result is produced in NC8 form only."""
    return (conv_w, stride, padding, padding_mode, conv_b)


def _tag_conv_output(ctx, pl, y, act):
    """_ConvFn.forward: mark an output whose gradient this layer's backward will read through NC8 kernels only (no fused
    activation: act_bwd reads the incoming gradient in NCHW).  A norm op told that it is the output's ONLY consumer
    (`private_input=True`) then hands its dx back in NC8 form alone."""
    if pl.bf16 and not ACT[act] and y.dtype == BF16:
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if _bwd_reads_only_nc8(pl, ctx.needs_input_grad[0], need_w):
            y._c2m_bwd_nc8 = True
    return y


def _virtual_grad(like, gn):
    """A gradient that exists in NC8 form only: a tensor object of the NCHW shape that carries `gn` for `_to_nc8` and whose own
    storage is never written or read (C2M_NC8_POISON=1 fills it with NaN -- the tests run the steps that way: a kernel that read
    it would poison the losses)."""
    g = torch.empty_like(like)
    if _NC8_POISON:
        g.fill_(float("nan"))
    return _mark_nc8_only(g, gn)


def _plan(x, w, stride, pad, reflect, dgrad_rows=None):
    key = (tuple(x.shape), tuple(w.shape), stride, pad, reflect, x.device.index, _conv_bf16, dgrad_rows)
    pl = _geom_cache.get(key)
    if pl is None:
        pl = _geom_cache[key] = _ConvPlan(tuple(x.shape), tuple(w.shape), stride, pad, reflect, x.device, _conv_bf16,
                                          dgrad_rows)
        if _conv_bf16:                     # geom[34] = operand precision, read by c2m_conv_igemm / c2m_conv_wgrad
            pl.fwd_geom[G.PRECISION] = pl.wg_geom[G.PRECISION] = 1
            for c in pl.classes:
                c["geom"][G.PRECISION] = 1
            for grp in (pl.cls_batch["groups"] if pl.cls_batch else ()):
                grp["geom"][G.PRECISION] = 1
    return pl


def _gp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _wino_filter(w, Cout, Cin, dgrad):
    """c2m_wino_filter_transform: native [Cout][Cin][3][3] -> packed U = G g G^T fragments (dgrad: transposed, rotated)."""
    L = _lib.lib()
    M, K = (Cin, Cout) if dgrad else (Cout, Cin)
    U = torch.empty(L.c2m_wino_upack_floats(M, K), device=w.device, dtype=torch.float32)
    _lib.check(L.c2m_wino_filter_transform(_p(w), _p(U), Cout, Cin, dgrad, _stream()), "wino_filter_transform")
    return U


def _pack_native(w, M, C, ck, kdims, stride, s_m, s_c):
    """c2m_pack_weights: contiguous native weights -> [ncls*M, nk*16] packed rows (ncls = prod(stride))."""
    kt, kh, kw = kdims
    st, sh, sw = stride
    ns = 16 // ck
    taps = (kt // st) * (kh // sh) * (kw // sw)
    out = torch.empty(st * sh * sw * M, _cdiv(C, ck) * _cdiv(taps, ns) * 16, device=w.device, dtype=torch.float32)
    g = np.array([M, C, ck, kt, kh, kw, st, sh, sw, s_m, s_c], dtype=np.int64)
    _lib.check(_lib.lib().c2m_pack_weights(_p(w), _p(out), _gp(g), _stream()), "pack_weights")
    out._c2m_job = (0, g)                   # how to rebuild this pack in place (refresh_trainable_packs)
    return out


_frozen_pack_cache = {}       # (id(w), kind) -> (weakref(w), w._version, w.data_ptr(), packed)   [name kept from round 1]
_PACK_CACHE_LIMIT = 4096


def _packed(w, frozen, kind, build):
    """Packed weight matrix (K-order rows, Winograd U fragments, bf16 patch image ...), built once per weight VALUE: cached
    while the SAME tensor object is alive and unmodified (weak reference + `_version` + data pointer: a new tensor that
    happens to reuse a freed allocation never hits).  Frozen weights (the VGG-19 of the perceptual loss) are packed once per
    run; trainable weights once per optimizer step -- `c2m_amd.optim.Adam.step` (and every in-place torch update) bumps
    `_version` -- instead of once per call: the discriminators run three times per step, every layer's forward and data
    gradient layouts were rebuilt on every use (pack_weights 76 + wino_filter 94 launches per step in round 2).
    Under a HIP-graph capture trainable weights are re-packed INSIDE the graph and never cached: the graph is replayed after
    eager optimizer steps, and a pack tensor from the graph's private pool must not leak into eager code."""
    key = (id(w), kind)
    hit = _frozen_pack_cache.get(key)
    fresh = hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr()
    if not frozen and w.is_cuda and torch.cuda.is_current_stream_capturing():
        # a pack that is kept fresh from outside the graph (refresh_trainable_packs after every optimizer step: same buffer,
        # new contents) is used as it is -- the graph then holds no pack kernels; pinned, because the graph keeps its address
        if _PACK_REFRESH and fresh and key in _pack_jobs and _pack_jobs[key][3] is hit[3]:
            _capture_pins[id(hit[3])] = hit[3]
            return hit[3]
        return build()
    if fresh:
        return hit[3]
    # A registered pack of this very tensor that has gone stale outside the optimizer (load_state_dict, .copy_, a torch
    # optimizer): rebuild it IN PLACE.  A captured graph may hold the buffer's address (_capture_pins); a fresh allocation here
    # would leave that graph replaying the old weights for ever.
    job = _pack_jobs.get(key)
    if _PACK_REFRESH and job is not None and hit is not None and job[0]() is w and hit[0]() is w and job[3] is hit[3] and \
            hit[2] == w.data_ptr():
        refresh_trainable_packs([w])
        return job[3]
    if len(_frozen_pack_cache) > _PACK_CACHE_LIMIT:
        for k in [k for k, v in _frozen_pack_cache.items() if v[0]() is None]:       # owners that died (e.g. the
            del _frozen_pack_cache[k]                                                  # per-forward spectral-norm weight)
        if len(_frozen_pack_cache) > _PACK_CACHE_LIMIT:
            _frozen_pack_cache.clear()
    A = build()
    _frozen_pack_cache[key] = (weakref.ref(w), w._version, w.data_ptr(), A)
    job = getattr(A, "_c2m_job", None)
    # (leaf tensors only: a weight derived per forward -- spectral norm's w / sigma -- dies with its step, and registering it would
    # throw the device tables away every step: +2-3 ms of host time per eager step when that was tried)
    if not frozen and job is not None and w.is_leaf and A.data_ptr() != w.data_ptr():
        _pack_jobs[key] = (weakref.ref(w), job[0], job[1], A)
        _pack_tables.clear()          # device job tables are keyed on (key, pointers): a new tensor may reuse both with another geometry
    return A


_PACK_REFRESH = os.environ.get("C2M_PACK_REFRESH", "1") != "0"      # A/B knob: 0 = packs rebuilt lazily, one launch each
_pack_jobs = {}          # (id(w), kind) -> (weakref(w), job type, g, packed tensor): packs of trainable weights we know how to rebuild
_pack_tables = {}        # signature of a stale set -> (device job table, njobs, total workgroups)
_capture_pins = {}       # id -> pack whose address a captured graph holds (one entry per pack, however many captures)


def refresh_trainable_packs(params=None):
    """Rebuild, in ONE launch (c2m_pack_multi), every cached pack of a trainable weight whose weight has changed since it was
    packed -- called by c2m_amd.optim.Adam.step and before a HIP-graph replay.  The packs keep their buffers, so the ~195 pack
    launches a full G + D step used to spend on its first use of every layer after the optimizer step (1.3 ms of GPU time, more
    host time than that) become one, and a captured graph needs no pack kernels at all.  Returns the number of packs rebuilt."""
    if not _PACK_REFRESH or not _pack_jobs:
        return 0
    stale = []
    if params is None:
        items = list(_pack_jobs.items())
    else:                                   # the optimizer's own parameters only (host time: ~1.5 us per registered pack)
        ids = {id(p) for p in params}
        items = [(k, v) for k, v in _pack_jobs.items() if k[0] in ids]
    for key, (wref, typ, g, A) in items:
        w = wref()
        hit = _frozen_pack_cache.get(key)
        if w is None or hit is None or hit[3] is not A or hit[0]() is not w or hit[2] != w.data_ptr():
            del _pack_jobs[key]
            continue
        if hit[1] != w._version:
            stale.append((key, w, typ, g, A))
    if not stale:
        return 0
    L = _lib.lib()
    sig = tuple(id(A) for _, _, _, _, A in stale)         # packs stay alive while registered; tables die with any new registration
    tab = _pack_tables.get(sig)
    if tab is None:
        nb = L.c2m_pack_job_bytes()
        host = (ctypes.c_ubyte * (nb * len(stale)))()
        first = 0
        btab = []
        for i, (_, w, typ, g, A) in enumerate(stale):
            n = L.c2m_pack_job_fill(ctypes.addressof(host) + i * nb, typ, _p(w), _p(A), _gp(g), first)
            if n <= 0:
                raise RuntimeError("c2m_pack_job_fill: bad pack geometry")
            first += n
            btab.append(np.stack([np.full(n, i, dtype=np.int32), np.arange(n, dtype=np.int32)], 1))
        device = stale[0][1].device
        dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        bdev = torch.from_numpy(np.concatenate(btab, 0)).to(device)
        if len(_pack_tables) > 16:
            _pack_tables.clear()
        tab = _pack_tables[sig] = (dev, len(stale), first, bdev)
    _lib.check(L.c2m_pack_multi(_p(tab[0]), _p(tab[3]), tab[1], tab[2], _stream()), "pack_multi")
    for key, w, _, _, A in stale:
        hit = _frozen_pack_cache[key]
        _frozen_pack_cache[key] = (hit[0], w._version, hit[2], A)
    return len(stale)


def _set_io(geom, x, ydt):
    """Per-call element types / bounds of a c2m_conv_igemm geom: X bytes (buffer range check), X type, Y type."""
    geom[G.X_BYTES] = x.numel() * x.element_size()
    geom[G.X_TYPE] = _dt(x)
    geom[G.Y_TYPE] = 1 if ydt == BF16 else 0
    return geom


def _conv_dgrad(pl, w, gy, frozen_w, out_dtype=torch.float32, keep=None):
    """Data gradient of the convolution described by plan `pl` (= the transposed convolution of gy with w): shared by
    _ConvFn.backward and conv_transpose2d.  gy contiguous [N, Cout, (To,) Ho, Wo]; returns [N, Cin, (Ti,) Hi, Wi] in
    `out_dtype` (the dtype of the forward input: bf16 on the bf16 data path, fp32 for fp32 inputs)."""
    if not pl.bf16:
        gy = _as(gy, torch.float32)
    L = _lib.lib()
    N, Cin, Cout = pl.dims[0:3]
    Ti_, Hi_, Wi_ = pl.dims[3:6]
    xshape = (N, Cin, Ti_, Hi_, Wi_) if pl.is3d else (N, Cin, Hi_, Wi_)
    xnumel = N * Cin * Ti_ * Hi_ * Wi_
    dev = gy.device
    gx = None
    if pl.wino_dgrad and pl.wino3d:
        dM = pl.dM
        # virtual channels (flipped time tap, output channel): a "native" [3*Cout][dM][3][3] weight for the 2-D transform
        U = _packed(w, frozen_w, ("wino-dgrad3d", dM), lambda: _wino_filter(
            w[:, :dM].flip(2).permute(2, 0, 1, 3, 4).reshape(3 * Cout, dM, 3, 3).contiguous(), 3 * Cout, dM, 1))
        gx = torch.empty(xshape, device=dev, dtype=torch.float32)
        tgt = torch.empty(pl.wino_dgrad_target, device=dev, dtype=torch.float32) if pl.reflect else gx
        g3 = pl.wino_dgrad_geom
        npix = int(g3[WG.NIMG] * g3[WG.HO] * g3[WG.WO])
        tag = ("dgrad", Cin, Cout * 27, npix, pl.dims[9:12], pl.stride, pl.reflect, "wino")
        two = pl.reflect and pl.wino3d_pairs
        _lib.check(_timed("wino", pl.dgrad_flops,
                          lambda: L.c2m_conv_wino(_p(U), _p(gy), _p(tgt), _p(gx) if two else None, None, _gp(g3), 0, 0.0,
                                                  _stream()), tag,
                          4 * (gy.numel() + w.numel() + xnumel)), "conv_wino dgrad 3-D")
        if pl.reflect:
            Ti, Hi, Wi = pl.dims[3:6]
            if two:
                _lib.check(L.c2m_reflect_border_add(_p(tgt), _p(gx), N * Cin, Ti, Hi, Wi, 0, 1, 1, 0, _stream()),
                           "reflect border add 3-D")
            else:
                _lib.check(L.c2m_reflect_fold(_p(tgt), _p(gx), N * Cin, Ti, Hi, Wi, 1, 1, 1, 0, _stream()), "reflect fold 3-D")
        if dM < Cin:
            gx[:, dM:].zero_()
    elif pl.wino_dgrad:
        w4 = pl.wino4_dgrad
        U = _packed(w, frozen_w, ("wino4-dgrad",), lambda: _wino4_filter(w, Cout, Cin, 1)) if w4 else \
            _packed(w, frozen_w, ("wino-dgrad",), lambda: _wino_filter(w, Cout, Cin, 1))
        conv_wino = L.c2m_conv_wino4 if w4 else L.c2m_conv_wino
        gx = torch.empty(xshape, device=dev, dtype=torch.float32)
        npix = int(pl.wino_dgrad_geom[WG.NIMG] * pl.wino_dgrad_geom[WG.HO] * pl.wino_dgrad_geom[WG.WO])
        tag = ("dgrad", Cin, Cout * 9, npix, pl.dims[9:12], pl.stride, pl.reflect, "wino4" if w4 else "wino")
        if pl.ring_dgrad:
            # reflect, exact domain: the interior term straight into gx, then the pad ring added in place (conv_ring.hip) -- both
            # launches inside the timed region of this layer's data gradient
            Ar = _packed(w, frozen_w, ("ring-dgrad",), lambda: _ring_pack(w, Cout, Cin))
            Hi_, Wi_ = pl.dims[4:6]
            g_ = pl.wino_dgrad_geom
            if _RING_BUFFER:
                # buffer form: the ring launch WRITES its terms (coalesced, no read-modify-write of dX, no corner part) and the
                # Winograd launch behind it adds them in its epilogue
                r_l = _ceil(max(Hi_, Wi_), 4)
                R = torch.empty(N * Cin * 4 * r_l, device=dev, dtype=torch.float32)

                def run_ring():
                    rc = L.c2m_reflect_ring_buffer(_p(Ar), _p(gy), _p(R), N, Cout, Cin, Hi_, Wi_, r_l, _stream())
                    g_[WG.RING], g_[WG.RING_L] = R.data_ptr(), r_l
                    try:
                        return rc or conv_wino(_p(U), _p(gy), _p(gx), None, None, _gp(g_), 0, 0.0, _stream())
                    finally:
                        g_[WG.RING], g_[WG.RING_L] = 0, 0
            else:
                def run_ring():
                    rc = conv_wino(_p(U), _p(gy), _p(gx), None, None, _gp(g_), 0, 0.0, _stream())
                    return rc or L.c2m_reflect_ring_dgrad(_p(Ar), _p(w), _p(gy), _p(gx), N, Cout, Cin, Hi_, Wi_, _stream())
            _lib.check(_timed("wino4" if w4 else "wino", pl.fwd_flops, run_ring, tag + ("ring",),
                              4 * (gy.numel() + w.numel() + xnumel)), "conv_wino dgrad + reflect ring")
            return gx if gx.dtype == out_dtype else gx.to(out_dtype)
        # reflect: ring of the padded domain -> tgt (only the ring is ever written or read), interior -> gx
        tgt = torch.empty(pl.dgrad_target, device=dev, dtype=torch.float32) if pl.reflect else gx
        _lib.check(_timed("wino4" if w4 else "wino", pl.fwd_flops,
                          lambda: conv_wino(_p(U), _p(gy), _p(tgt), _p(gx) if pl.reflect else None, None,
                                            _gp(pl.wino_dgrad_geom), 0, 0.0, _stream()), tag,
                          4 * (gy.numel() + w.numel() + xnumel)), "conv_wino dgrad")
        if pl.reflect:
            Ti, Hi, Wi = pl.dims[3:6]
            _lib.check(L.c2m_reflect_border_add(_p(tgt), _p(gx), N * Cin, Ti, Hi, Wi, 0, 1, 1, 0, _stream()),
                       "reflect border add")
    elif pl.bf16 and pl.k333_dgrad_nc8:
        # 3x3x3 layers: (sample, frame) images, frame t summing its (dY frame, time tap) pairs; reflect: spatially padded target + fold
        gy_b = _as(gy, BF16)
        kdt = BF16 if out_dtype == BF16 else torch.float32
        dM = pl.dM
        A = _packed(w, frozen_w, ("dgrad-bf16-k333", dM), lambda: _pack_bf16_k333(w, dM, Cout, dgrad=True, cin_total=Cin))
        gx = torch.empty(xshape, device=dev, dtype=kdt)
        tgt = torch.empty((N, Cin, Ti_, Hi_ + 2, Wi_ + 2), device=dev, dtype=kdt) if pl.reflect else gx
        tag = ("dgrad", Cin, Cout * 27, int(N * Ti_ * Hi_ * Wi_), pl.dims[9:12], pl.stride, pl.reflect, "nc8")

        def run_k333d():
            gyn = _to_nc8(gy_b, keep)
            return L.c2m_conv3d_dgrad_nc8(_p(A), _p(gyn), _p(tgt), _p(pl.k333_ptab), dM, Cin, Cout, N, Ti_, Hi_, Wi_, int(pl.reflect),
                                          _dt(tgt), _stream())
        _lib.check(_timed("igemm_bf16", pl.dgrad_flops, run_k333d, tag, 2 * (gy.numel() + xnumel) + 4 * w.numel()), "conv3d_dgrad_nc8")
        if pl.reflect:
            _lib.check(L.c2m_reflect_fold(_p(tgt), _p(gx), N * Cin, Ti_, Hi_, Wi_, 0, 1, 1, _dt(tgt), _stream()), "reflect fold (3-D, spatial)")
        if dM < Cin:
            gx[:, dM:].zero_()               # channels declared gradient-free by the caller (dgrad_channels); their planes of tgt were never written
    elif pl.bf16 and pl.s2_dgrad_nc8:
        # 4x4 stride-2 layers: all four output parity classes in one launch on the NC8 form of dY (conv_nc8.hip)
        gy_b = _as(gy, BF16)
        Ho_, Wo_ = pl.dims[7:9]
        kdt = BF16 if out_dtype == BF16 else torch.float32
        A = _packed(w, frozen_w, ("dgrad-bf16-s2", pl.reflect), lambda: _pack_bf16_patch(w, Cin, Cout, 16, Cin * 16, 4 if pl.reflect else 3))
        gx = torch.empty(xshape, device=dev, dtype=kdt)
        tgt = torch.empty(pl.dgrad_target, device=dev, dtype=kdt) if pl.reflect else gx
        tag = ("dgrad", Cin, Cout * 16, int(N * Hi_ * Wi_), pl.dims[9:12], pl.stride, pl.reflect, "nc8")

        def run_s2d():
            gyn = _to_nc8(gy_b, keep)
            return L.c2m_conv_s2_dgrad_nc8(_p(A), _p(gyn), _p(tgt), Cin, Cout, N, Ho_, Wo_, int(pl.reflect), _dt(tgt), _stream())
        _lib.check(_timed("igemm_bf16", pl.dgrad_flops, run_s2d, tag, 2 * (gy.numel() + xnumel) + 4 * w.numel()), "conv_s2_dgrad_nc8")
        if pl.reflect:
            _lib.check(L.c2m_reflect_fold(_p(tgt), _p(gx), N * Cin, 1, Hi_, Wi_, 0, 1, 1, _dt(tgt), _stream()), "reflect fold (s2)")
    else:
        g8 = pl.bf16 and pl.g8_dgrad                 # the NC8 gather form of the same launches (conv_gather_nc8_kernel)
        S = pl.g8_dgrad_splits if g8 else pl.dgrad_splits
        folded = pl.reflect and any(pl.pad)
        two_target = folded and S == 1 and not pl.dgrad_needs_zero
        alloc = torch.zeros if pl.dgrad_needs_zero else torch.empty
        cb = pl.cls_batch
        # bf16 data path: the bf16 kernels gather gy as bf16 and write `kdt`; the <= 4-row vector-ALU kernels (dgrad of a
        # <= 4-channel input) are fp32 in, fp32 out -- if any launch of this layer is one of those, the layer's target is fp32
        gy_b = gy_f = None
        thin_of = {}
        kdt = torch.float32
        if pl.bf16:
            if cb is None:
                thin_of = {id(c): _thin(pl.dM, S, c["npix"], two_target, 1) for c in pl.classes}
            any_thin = any(thin_of.values())
            kdt = BF16 if (out_dtype == BF16 and not any_thin) else torch.float32
            if cb is not None or not all(thin_of.values()):
                gy_b = _as(gy, BF16)
            if any_thin:
                gy_f = _as(gy, torch.float32)
        tgt = alloc(pl.dgrad_target, device=dev, dtype=kdt)
        gx = torch.empty(xshape, device=dev, dtype=kdt) if folded else tgt.view(xshape)
        dst = tgt if S == 1 else alloc(S * tgt.numel(), device=dev, dtype=torch.float32)
        st, sh, sw = pl.stride
        w5 = w if pl.is3d else w.unsqueeze(2)
        if g8:
            kt, kh, kw = pl.dims[9:12]
            A8 = _packed(w, frozen_w, ("dgrad-bf16-g8", pl.dM, pl.stride), lambda: _pack_bf16_gather(
                w, pl.dM, Cout, (kt, kh, kw), pl.stride, kt * kh * kw, Cin * kt * kh * kw))
            launches = [(grp["g8"], grp["g8tab"], grp["first"], cb["taps"] * grp["npix"] * grp["ncls"], grp["npix"] * grp["ncls"])
                        for grp in cb["groups"]] if cb is not None else \
                       [(c["g8"], c["g8tab"], ci, c["taps"] * c["npix"], c["npix"]) for ci, c in enumerate(pl.classes)]
            for geom8, tab8, first, work, npix in launches:
                tag = ("dgrad", Cin, Cout * pl.classes[0]["taps"], npix, pl.dims[9:12], pl.stride, pl.reflect, S, "nc8g")
                Ag = A8[first * pl.g8_a_cls:]

                def run_g8d():
                    gyn = _to_nc8(gy_b, keep)
                    _set_io(geom8, gyn, kdt)
                    return L.c2m_conv_igemm(_p(Ag), _p(gyn), _p(dst), _p(gx) if two_target else None, None, _p(tab8), _gp(geom8),
                                            0, 0.0, _stream())
                _lib.check(_timed("igemm_bf16", pl.dgrad_flops * work / pl.dgrad_work, run_g8d, tag,
                                  (2 * (gy.numel() + xnumel) + 4 * w.numel()) * work // int(pl.dgrad_work)),
                           "conv_igemm dgrad (NC8 gather)")
        elif cb is not None:
            kt, kh, kw = pl.dims[9:12]
            A = _packed(w, frozen_w, ("dgrad-all", cb["ck"], pl.stride), lambda: _pack_native(
                w, Cin, Cout, cb["ck"], (kt, kh, kw), pl.stride, kt * kh * kw, Cin * kt * kh * kw))
            for grp in cb["groups"]:
                Ag = A[grp["first"] * Cin:]
                tag = ("dgrad", Cin, Cout * cb["taps"], grp["npix"] * grp["ncls"], pl.dims[9:12], pl.stride,
                       pl.reflect, S)
                gin = gy_b if pl.bf16 else gy
                _set_io(grp["geom"], gin, kdt)
                _lib.check(_timed("igemm_bf16" if pl.bf16 else "igemm",
                                  pl.dgrad_flops * cb["taps"] * grp["npix"] * grp["ncls"] / pl.dgrad_work,
                                  lambda: L.c2m_conv_igemm(_p(Ag), _p(gin), _p(dst), _p(gx) if two_target else None,
                                                           None, _p(grp["tab"]), _gp(grp["geom"]), 0, 0.0,
                                                           _stream()), tag,
                                  4 * (gy.numel() * grp["ncls"] // cb["ncls"] + w.numel() +
                                       xnumel * grp["ncls"] // cb["ncls"])),
                           "conv_igemm dgrad (batched classes)")
        kt, kh, kw = pl.dims[9:12]
        Aall = None
        if cb is None and not g8 and pl.classes_packable and not (pl.bf16 and all(c["patch"] for c in pl.classes)):
            Aall = _packed(w, frozen_w, ("dgrad-all", pl.classes[0]["ck"], pl.stride), lambda: _pack_native(
                w, Cin, Cout, pl.classes[0]["ck"], (kt, kh, kw), pl.stride, kt * kh * kw, Cin * kt * kh * kw))
        for ci, c in enumerate(pl.classes if (cb is None and not g8) else ()):
            rt, ry, rx = c["r"]
            A = _packed(w, frozen_w, ("dgrad-bf16-patch", pl.dM), lambda: _pack_bf16_patch(
                w, pl.dM, Cout, 9, Cin * 9)) if (c["patch"] and pl.bf16) else \
                Aall[ci * Cin:] if Aall is not None else _packed(
                w, frozen_w, ("dgrad", c["ck"], pl.stride, c["r"]), lambda: _pack_rows(
                    w5[:, :, rt::st, ry::sh, rx::sw].reshape(Cout, Cin, c["taps"]).transpose(0, 1), c["ck"]))
            tag = ("dgrad", Cin, Cout * c["taps"], c["npix"], pl.dims[9:12], pl.stride, pl.reflect, S)
            gin = (gy_f if thin_of.get(id(c)) else gy_b) if pl.bf16 else gy
            _set_io(c["geom"], gin, torch.float32 if thin_of.get(id(c)) else kdt)
            nc8 = c["patch"] and pl.bf16 and _NC8 and not pl.is3d and (pl.dims[7] * pl.dims[8]) % 8 == 0
            _lib.check(_timed("igemm_bf16" if pl.bf16 else "igemm", pl.dgrad_flops * c["taps"] * c["npix"] / pl.dgrad_work,
                              (lambda: _nc8_launch(L, A, gin, dst, gx if two_target else None, None, c["geom"], 0, 0.0, keep)) if nc8 else
                              (lambda: L.c2m_conv_igemm(_p(A), _p(gin), _p(dst), _p(gx) if two_target else None, None,
                                                        _p(c["tab"]), _gp(c["geom"]), 0, 0.0, _stream())),
                              tag + (("nc8",) if nc8 else ()),
                              4 * (gy.numel() + w.numel() + xnumel) // len(pl.classes)), "conv_igemm dgrad")
        if S > 1:
            _lib.check(L.c2m_splitk_reduce(_p(dst), _p(tgt), None, tgt.numel(), S, 1, 1, 0, 0.0, _dt(tgt), _stream()),
                       "splitk_reduce dgrad")
        if folded:
            Ti, Hi, Wi = pl.dims[3:6]
            fold = L.c2m_reflect_border_add if two_target else L.c2m_reflect_fold
            _lib.check(fold(_p(tgt), _p(gx), N * Cin, Ti, Hi, Wi, pl.pad[0], pl.pad[1], pl.pad[2], _dt(tgt), _stream()),
                       "reflect fold")
        if pl.dM < Cin:
            gx[:, pl.dM:].zero_()            # channels declared gradient-free by the caller (dgrad_rows)
    return gx if gx.dtype == out_dtype else gx.to(out_dtype)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, reflect, act, dgrad_rows=None, slope=LRELU_SLOPE):
        _dev(x, w, b)
        ctx.slope = slope = float(slope)
        x, w = _f(x), _f(w)
        pl = _plan(x, w, stride, pad, reflect, dgrad_rows)
        L = _lib.lib()
        N, Cin, Cout = pl.dims[0:3]
        if b is not None and b.numel() != Cout:
            raise RuntimeError(f"convolution: bias of {b.numel()} elements for {Cout} output channels")
        ctx.frozen_w = not ctx.needs_input_grad[1]
        ctx.x_dtype = x.dtype
        ctx.nc8_keep = None
        if not pl.bf16:
            x = _as(x, torch.float32)            # the fp32 kernels are fp32 in, fp32 out
        if pl.wino_fwd:
            if pl.wino3d:      # virtual channels (kt, ci): [Cout][3*Cin][3][3]
                U = _packed(w, ctx.frozen_w, ("wino-fwd3d",), lambda: _wino_filter(
                    w.permute(0, 2, 1, 3, 4).reshape(Cout, 3 * Cin, 3, 3).contiguous(), Cout, 3 * Cin, 0))
            elif pl.wino4_fwd:
                U = _packed(w, ctx.frozen_w, ("wino4-fwd",), lambda: _wino4_filter(w, Cout, Cin, 0))
            else:
                U = _packed(w, ctx.frozen_w, ("wino-fwd",), lambda: _wino_filter(w, Cout, Cin, 0))
            conv_wino = L.c2m_conv_wino4 if (pl.wino4_fwd and not pl.wino3d) else L.c2m_conv_wino
            y = torch.empty(pl.out_shape, device=x.device, dtype=torch.float32)
            w4 = pl.wino4_fwd and not pl.wino3d           # F(4x4,3x3): its own roofline family (executed = algorithmic / 4)
            tag = ("fwd", Cout, pl.K, int(pl.fwd_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "wino4" if w4 else "wino")
            _lib.check(_timed("wino4" if w4 else "wino", 2.0 * Cout * pl.K * int(pl.fwd_geom[G.NPIX]),
                              lambda: conv_wino(_p(U), _p(x), _p(y), None, _p(b), _gp(pl.wino_fwd_geom), ACT[act],
                                                slope, _stream()), tag,
                              4 * (x.numel() + w.numel() + y.numel())), "conv_wino fwd")
            ctx.pl, ctx.act, ctx.has_bias = pl, act, b is not None
            ctx.save_for_backward(x, w, y if ACT[act] else None)
            return _tag_conv_output(ctx, pl, y, act)
        if pl.bf16 and pl.k333_nc8:
            x = _as(x, BF16)
            A = _packed(w, ctx.frozen_w, ("fwd-bf16-k333",), lambda: _pack_bf16_k333(w, Cout, Cin))
            y = torch.empty(pl.out_shape, device=x.device, dtype=BF16)
            Ti_, Hi_, Wi_ = pl.dims[3:6]
            tag = ("fwd", Cout, pl.K, int(pl.fwd_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "nc8")
            ctx.nc8_keep = {} if (pl.k333_wgrad_nc8 and ctx.needs_input_grad[1]) else None

            def run_k333():
                xn = _to_nc8(x, ctx.nc8_keep)
                return L.c2m_conv3d_nc8(_p(A), _p(xn), _p(y), _p(b), Cout, Cin, N, Ti_, Hi_, Wi_, int(pl.reflect), 1, ACT[act], slope,
                                        _stream())
            _lib.check(_timed("igemm_bf16", 2.0 * Cout * pl.K * int(pl.fwd_geom[G.NPIX]), run_k333, tag,
                              2 * (x.numel() + y.numel()) + 4 * w.numel()), "conv3d_nc8 fwd")
            ctx.pl, ctx.act, ctx.has_bias = pl, act, b is not None
            ctx.save_for_backward(x, w, y if ACT[act] else None)
            return _tag_conv_output(ctx, pl, y, act)
        if pl.bf16 and pl.s2_nc8:
            x = _as(x, BF16)
            A = _packed(w, ctx.frozen_w, ("fwd-bf16-s2",), lambda: _pack_bf16_patch(w, Cout, Cin, pl.K, 16, 2))
            y = torch.empty(pl.out_shape, device=x.device, dtype=BF16)
            Hi_, Wi_ = pl.dims[4:6]
            tag = ("fwd", Cout, pl.K, int(pl.fwd_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "nc8")

            ctx.nc8_keep = {} if (pl.s2_wgrad_nc8 and ctx.needs_input_grad[1]) else None

            def run_s2():
                xn = _to_nc8(x, ctx.nc8_keep)
                return L.c2m_conv_s2_nc8(_p(A), _p(xn), _p(y), _p(b), Cout, Cin, N, Hi_, Wi_, int(pl.reflect), 1, ACT[act], slope,
                                         _stream())
            _lib.check(_timed("igemm_bf16", 2.0 * Cout * pl.K * int(pl.fwd_geom[G.NPIX]), run_s2, tag,
                              2 * (x.numel() + y.numel()) + 4 * w.numel()), "conv_s2_nc8 fwd")
            ctx.pl, ctx.act, ctx.has_bias = pl, act, b is not None
            ctx.save_for_backward(x, w, y if ACT[act] else None)
            return _tag_conv_output(ctx, pl, y, act)
        if pl.bf16 and pl.g8_fwd:
            # the NC8 gather form: channel-blocked input, bf16 weights in (tap, chunk) order (conv_gather_nc8_kernel)
            x = _as(x, BF16)
            A = _packed(w, ctx.frozen_w, ("fwd-bf16-g8",), lambda: _pack_bf16_gather(w, Cout, Cin, pl.dims[9:12], (1, 1, 1), pl.K, pl.K // Cin))
            S = pl.g8_fwd_splits
            y = torch.empty(pl.out_shape, device=x.device, dtype=BF16)
            dst = y if S == 1 else torch.empty(S * y.numel(), device=x.device, dtype=torch.float32)
            tag = ("fwd", Cout, pl.K, int(pl.fwd_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, S, "nc8g")

            def run_g8():
                xn = _to_nc8(x)
                _set_io(pl.g8_fwd_geom, xn, BF16)
                return L.c2m_conv_igemm(_p(A), _p(xn), _p(dst), None, _p(b), _p(pl.g8_fwd_tab), _gp(pl.g8_fwd_geom), ACT[act], slope,
                                        _stream())
            _lib.check(_timed("igemm_bf16", 2.0 * Cout * pl.K * int(pl.fwd_geom[G.NPIX]), run_g8, tag,
                              2 * (x.numel() + y.numel()) + 4 * w.numel()), "conv_igemm fwd (NC8 gather)")
            if S > 1:
                _lib.check(L.c2m_splitk_reduce(_p(dst), _p(y), _p(b), y.numel(), S, int(pl.fwd_geom[G.OUT_SC]), Cout, ACT[act],
                                               slope, _dt(y), _stream()), "splitk_reduce")
            ctx.pl, ctx.act, ctx.has_bias = pl, act, b is not None
            ctx.save_for_backward(x, w, y if ACT[act] else None)
            return _tag_conv_output(ctx, pl, y, act)
        if pl.fwd_patch and pl.bf16:
            A = _packed(w, ctx.frozen_w, ("fwd-bf16-patch",), lambda: _pack_bf16_patch(w, Cout, Cin, pl.K, 9))
        else:
            A = _packed(w, ctx.frozen_w, ("fwd", pl.ck), lambda: _pack_native(w, Cout, Cin, pl.ck, pl.dims[9:12], (1, 1, 1), pl.K, pl.K // Cin))
        S = pl.fwd_splits
        ydt = torch.float32
        if pl.bf16:
            # bf16 data path: bf16 in (an fp32 input is cast once), bf16 out -- except the <= 4-channel heads (flow, occlusion,
            # RGB), which stay fp32 and run on the fp32 vector-ALU kernels when the launch is big enough for them
            if _thin(Cout, S, int(pl.fwd_geom[G.NPIX])):
                x = _as(x, torch.float32)
            else:
                x = _as(x, BF16)
                ydt = BF16 if Cout > 4 else torch.float32
        y = torch.empty(pl.out_shape, device=x.device, dtype=ydt)
        dst = y if S == 1 else torch.empty(S * y.numel(), device=x.device, dtype=torch.float32)
        tag = ("fwd", Cout, pl.K, int(pl.fwd_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, S)
        _set_io(pl.fwd_geom, x, ydt)
        nc8 = pl.fwd_patch and pl.bf16 and pl.nc8
        ctx.nc8_keep = {} if (nc8 and pl.wgrad_nc8 and ctx.needs_input_grad[1]) else None      # X in NC8 form, for the weight gradient
        _lib.check(_timed("igemm_bf16" if pl.bf16 else "igemm", 2.0 * Cout * pl.K * int(pl.fwd_geom[G.NPIX]),
                          (lambda: _nc8_launch(L, A, x, dst, None, b, pl.fwd_geom, ACT[act], slope, ctx.nc8_keep)) if nc8 else
                          (lambda: L.c2m_conv_igemm(_p(A), _p(x), _p(dst), None, _p(b), _p(pl.fwd_tab), _gp(pl.fwd_geom),
                                                    ACT[act], slope, _stream())), tag + (("nc8",) if nc8 else ()),
                          x.element_size() * x.numel() + 4 * w.numel() + y.element_size() * y.numel()), "conv_igemm fwd")
        if S > 1:
            _lib.check(L.c2m_splitk_reduce(_p(dst), _p(y), _p(b), y.numel(), S, int(pl.fwd_geom[G.OUT_SC]), Cout, ACT[act],
                                           slope, _dt(y), _stream()), "splitk_reduce")
        ctx.pl, ctx.act, ctx.has_bias = pl, act, b is not None
        ctx.save_for_backward(x, w, y if ACT[act] else None)
        return _tag_conv_output(ctx, pl, y, act)

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        pl, L = ctx.pl, _lib.lib()
        gy = _f(gy)
        if _NC8_LOG is not None:
            _NC8_LOG_TAG[0] = " (own act_bwd)" if ACT[ctx.act] else " (incoming gradient)"
        if ACT[ctx.act]:
            gy = _as(gy, y.dtype)
            need_w0 = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
            if y.dtype == BF16 and _bwd_reads_only_nc8(pl, ctx.needs_input_grad[0], need_w0):
                # the masked gradient in NC8 form only: its readers below are all NC8 kernels (no NCHW write, no layout pass)
                Nn, Cc = y.shape[0], y.shape[1]
                gn = torch.empty((Nn, _cdiv(Cc, 8)) + tuple(y.shape[2:]) + (8,), device=y.device, dtype=BF16)
                _lib.check(L.c2m_grad_to_nc8(0, _p(y), None, _p(gy), None, _p(gn), Nn, Cc, y.numel() // (Nn * Cc), 0, ACT[ctx.act],
                                             ctx.slope, _stream()), "grad_to_nc8 (act_bwd)")
                gy = _virtual_grad(y, gn)
            else:
                g = torch.empty_like(gy)
                _lib.check(L.c2m_act_bwd(_p(y), _p(gy), _p(g), gy.numel(), ACT[ctx.act], ctx.slope, _dt(y), _stream()), "act_bwd")
                gy = g
        N, Cin, Cout = pl.dims[0:3]
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        # The weight gradient and the data gradient of a layer are independent: with both wanted, the weight gradient (and its
        # split reductions) is issued on a side stream and joined at the end of this node, so the two launches share the chip
        # -- the tail of one (e.g. 1440 workgroups on 512 slots) is filled by the other and the ~10 us reductions disappear
        # behind MFMA kernels.  Fork / join are events, so a HIP-graph capture records the same parallel branches.
        # (inside a capture for both precisions: fp32 configs[1] as a replay 66.6 -> 66.1 ms on one box, `bench.py --graph`)
        side_on = _WGRAD_SIDE == "1" or (_WGRAD_SIDE == "auto" and torch.cuda.is_current_stream_capturing())
        # (not from the auxiliary stream of the object branch: its one small convolution gains nothing, and a side stream that two
        # captured streams fork into and join from takes capture_end down on ROCm 7.2 -- tools/aux_capture_probe.py)
        side = _side_stream(x.device) if (side_on and need_w and ctx.needs_input_grad[0] and not _on_aux_stream(x.device)) else None
        # NC8 form of dY: one layout pass shared by the data gradient and the weight gradient of this node (made on the main
        # stream BEFORE a fork, so the side stream's launch is ordered behind it)
        keep = {} if (pl.bf16 and (pl.nc8 or pl.k333_wgrad_nc8 or pl.k333_dgrad_nc8)) else None
        if keep is not None and need_w and (pl.wgrad_nc8 or pl.s2_wgrad_nc8 or pl.k333_wgrad_nc8):
            gy = _as(gy, BF16)
            _to_nc8(gy, keep)
        # (not while capturing a HIP graph: the graph executor runs the long side branch worse than the per-node fork + join below --
        # replays of configs[3] / configs[2]: 42.7 / 63.9 ms deferred, 42.4 / 63.5 per node, 43.8 / 66.4 deferred without the object
        # branch -- and a capture would have to hold every dY until its end)
        if need_w and _defer["on"] and not torch.cuda.is_current_stream_capturing():
            # `with ops.deferred_wgrads():` around backward (TrainStep): the weight gradient of a leaf weight that has no gradient
            # yet goes to the side stream and is NOT joined here -- nothing reads it before the optimizer (AccumulateGrad adopts the
            # tensor, no kernel) -- so its split reductions and its tail run under the data-gradient chain of the layers below.
            # The context's exit joins.  A weight that already has a gradient (a second backward) would be summed by a kernel on
            # this stream: it joins first and takes the ordinary path.  Limit of the scheme: a convolution weight that ALSO feeds a
            # non-convolution op in the same graph (none in this model) has the two gradients summed without that join -- run such a
            # model with C2M_DEFER_WGRAD=0.
            # (a weight applied twice in one graph -- the sparse-feature encoder with use_fw_of -- has its two gradients summed by
            # the engine before AccumulateGrad runs, on this stream: the second one joins as well)
            first_use = w.data_ptr() not in _defer["seen"]
            _defer["seen"].add(w.data_ptr())
            if first_use and w.is_leaf and w.grad is None and ConvProfiler.active is None and not _on_aux_stream(x.device):
                main = torch.cuda.current_stream(x.device)
                dside = _side_stream(x.device)
                dside.wait_stream(main)
                _record_stream_all(dside, x, gy, keep, ctx.nc8_keep)    # freed by autograd right after this node; still being read
                with torch.cuda.stream(dside):
                    gw, gb = _ConvFn._wgrad(ctx, pl, x, w, gy, keep)
                for t in (gw, gb):
                    if t is not None:
                        t.record_stream(main)
                        _defer["grads"][t.data_ptr()] = dside
                _defer["by_w"][w.data_ptr()] = [t.data_ptr() for t in (gw, gb) if t is not None]
                _defer["devs"].add(x.device.index)
                # dY may be shared with the identity path of a residual block (AddBackward hands ONE tensor to both branches): the
                # engine sums the other branch's gradient into it IN PLACE once nobody else holds it -- while the side stream still
                # reads it (found by test_branch_streams_do_not_change_a_step: conv2 of the residual blocks).  A second owner makes
                # that sum out of place; the reference is dropped when the side stream has passed this launch.
                ev = torch.cuda.Event()
                ev.record(dside)
                hold = _defer["hold"]
                hold.append((ev, gy))
                while hold and hold[0][0].query():
                    hold.popleft()
                if ctx.needs_input_grad[0]:
                    gx = _conv_dgrad(pl, w, gy, ctx.frozen_w, ctx.x_dtype, keep)
                return gx, gw, gb, None, None, None, None, None, None
            if x.device.index in _defer["devs"]:
                torch.cuda.current_stream(x.device).wait_stream(_side_stream(x.device))
                for ptr in _defer["by_w"].pop(w.data_ptr(), ()):     # its first gradient is summed with this one on THIS stream:
                    _defer["grads"].pop(ptr, None)                   # no longer "being produced on the side stream"
        if side is not None:
            main = torch.cuda.current_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gw, gb = _ConvFn._wgrad(ctx, pl, x, w, gy, keep)
            gx = _conv_dgrad(pl, w, gy, ctx.frozen_w, ctx.x_dtype, keep)
            main.wait_stream(side)
            for t in (gw, gb):
                if t is not None:
                    t.record_stream(main)       # allocated in the side stream's pool, consumed (and freed) on the main stream
            return gx, gw, gb, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            gx = _conv_dgrad(pl, w, gy, ctx.frozen_w, ctx.x_dtype, keep)
        if need_w:
            gw, gb = _ConvFn._wgrad(ctx, pl, x, w, gy, keep)
        return gx, gw, gb, None, None, None, None, None, None

    @staticmethod
    def _wgrad(ctx, pl, x, w, gy, keep=None):
        L = _lib.lib()
        N, Cin, Cout = pl.dims[0:3]
        gw = gb = None
        if not pl.bf16:
            gy = _as(gy, torch.float32)
        want = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if want and pl.bf16 and pl.k333_wgrad_nc8:
            Ti, Hi, Wi = pl.dims[3:6]
            gyn = _to_nc8(_as(gy, BF16), keep)
            xn = next(iter(ctx.nc8_keep.values()))[1] if ctx.nc8_keep else _to_nc8(_as(x, BF16))
            slab = torch.empty(L.c2m_conv_wgrad_nc8_slab_floats(Cout, Cin, N * Ti, Hi, Wi, 0), device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            gb_t = torch.empty(Cout, device=x.device, dtype=torch.float32)
            tag = ("wgrad", Cout, pl.K, int(pl.wg_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "nc8")
            _lib.check(_timed("wgrad_bf16", 2.0 * Cout * pl.K * int(pl.wg_geom[G.NPIX]),
                              lambda: L.c2m_conv_wgrad3d_nc8(_p(gyn), _p(xn), _p(slab), _p(gw), _p(gb_t), Cout, Cin, N, Ti, Hi, Wi,
                                                             int(pl.reflect), _stream()), tag,
                              2 * (gyn.numel() + xn.numel()) + 4 * w.numel()), "conv_wgrad3d_nc8")
            return gw, (gb_t if ctx.has_bias else None)
        if want and pl.bf16 and (pl.wgrad_nc8 or pl.s2_wgrad_nc8):
            # both operands in NC8 form: X from the forward launch (ctx.nc8_keep) or converted now, dY shared with the data gradient
            s2 = int(pl.s2_wgrad_nc8)
            Hi, Wi = pl.dims[7:9]                  # the dY map
            gyn = _to_nc8(_as(gy, BF16), keep)
            xn = next(iter(ctx.nc8_keep.values()))[1] if ctx.nc8_keep else _to_nc8(_as(x, BF16))
            slab = torch.empty(L.c2m_conv_wgrad_nc8_slab_floats(Cout, Cin, N, Hi, Wi, s2), device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            gb_t = torch.empty(Cout, device=x.device, dtype=torch.float32)
            tag = ("wgrad", Cout, pl.K, int(pl.wg_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "nc8")
            _lib.check(_timed("wgrad_bf16", 2.0 * Cout * pl.K * int(pl.wg_geom[G.NPIX]),
                              lambda: L.c2m_conv_wgrad_nc8(_p(gyn), _p(xn), _p(slab), _p(gw), _p(gb_t), Cout, Cin, N, Hi, Wi,
                                                           int(pl.reflect), s2, _stream()), tag,
                              2 * (gyn.numel() + xn.numel()) + 4 * w.numel()), "conv_wgrad_nc8")
            return gw, (gb_t if ctx.has_bias else None)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and pl.wino_wgrad and pl.wino_wgrad3d:
            S = pl.wino_wg_splits
            Ti, Hi, Wi = pl.dims[3:6]
            slab = torch.empty((S + 1) * 16 * Cout * 3 * Cin, device=x.device, dtype=torch.float32)
            dbslab = torch.empty(S * Cout, device=x.device, dtype=torch.float32)
            gw3 = torch.empty(Cout, 3, Cin, 3, 3, device=x.device, dtype=torch.float32)      # (time tap, channel) order
            gb_t = torch.empty(Cout, device=x.device, dtype=torch.float32)
            tag = ("wgrad", Cout, pl.K, int(pl.wg_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "wino")
            _lib.check(_timed("wino_wgrad", 2.0 * Cout * pl.K * int(pl.wg_geom[G.NPIX]),
                              lambda: L.c2m_conv_wino_wgrad3d(_p(gy), _p(x), _p(slab), _p(dbslab), _p(gw3), _p(gb_t), Cout,
                                                              Cin, N, Ti, Hi, Wi, int(pl.reflect), _stream()), tag,
                              4 * (gy.numel() + x.numel() + w.numel())), "conv_wino_wgrad3d")
            gw = gw3.permute(0, 2, 1, 3, 4).contiguous()
            gb = gb_t if ctx.has_bias else None
        elif (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and pl.wino_wgrad:
            S = pl.wino_wg_splits
            slab = torch.empty((S + 1) * 16 * Cout * Cin, device=x.device, dtype=torch.float32)
            dbslab = torch.empty(S * Cout, device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            gb_t = torch.empty(Cout, device=x.device, dtype=torch.float32)
            Hi, Wi = pl.dims[4:6]
            tag = ("wgrad", Cout, pl.K, int(pl.wg_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, "wino")
            _lib.check(_timed("wino_wgrad", 2.0 * Cout * pl.K * int(pl.wg_geom[G.NPIX]),
                              lambda: L.c2m_conv_wino_wgrad(_p(gy), _p(x), _p(slab), _p(dbslab), _p(gw), _p(gb_t), Cout,
                                                            Cin, N, Hi, Wi, int(pl.reflect), _stream()), tag,
                              4 * (gy.numel() + x.numel() + w.numel())), "conv_wino_wgrad")
            gb = gb_t if ctx.has_bias else None
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            slab = torch.empty(pl.wg_splits * Cout * pl.J, device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            gb_t = torch.empty(Cout, device=x.device, dtype=torch.float32)
            tag = ("wgrad", Cout, pl.K, int(pl.wg_geom[G.NPIX]), pl.dims[9:12], pl.stride, pl.reflect, pl.wg_splits)
            # bf16 data path: the MFMA kernel gathers bf16 dY and X; the <= 4-output-channel heads run on the fp32 vector-ALU
            # kernels (c2m_conv_wgrad's own rule: M <= 4 and >= 16384 pixels)
            wdt = BF16 if (pl.bf16 and not (Cout <= 4 and int(pl.wg_geom[G.NPIX]) >= 16384)) else torch.float32
            xg, gyw = _as(x, wdt), _as(gy, wdt)
            pl.wg_geom[G.X_BYTES], pl.wg_geom[G.DY_BYTES] = xg.numel() * xg.element_size(), gyw.numel() * gyw.element_size()
            pl.wg_geom[G.X_TYPE] = _dt(xg)
            _lib.check(_timed("wgrad_bf16" if pl.bf16 else "wgrad", 2.0 * Cout * pl.K * int(pl.wg_geom[G.NPIX]),
                              lambda: L.c2m_conv_wgrad(_p(gyw), _p(xg), _p(slab), _p(gw), _p(gb_t), _p(pl.wg_tab),
                                                       _gp(pl.wg_geom), _stream()), tag,
                              xg.element_size() * (gyw.numel() + xg.numel()) + 4 * w.numel()), "conv_wgrad")
            gb = gb_t if ctx.has_bias else None
        return gw, gb


def conv(x, w, b=None, stride=1, padding=0, padding_mode="zeros", act=None, dgrad_channels=None, slope=LRELU_SLOPE):
    """conv2d (4-D x) / conv3d (5-D x) with zero or reflect padding folded into the gather; bias + activation fused.
    dgrad_channels: the caller guarantees that only x[:, :dgrad_channels] needs a gradient (x is a concatenation whose tail
    carries no grad); the data gradient of the tail is returned as zeros instead of being computed."""
    nd = x.dim() - 2
    stride3, pad3 = _triple(stride, nd), _pad3(padding, nd)
    reflect = padding_mode == "reflect" and any(pad3)
    if padding_mode not in ("zeros", "reflect"):
        raise NotImplementedError(f"padding_mode {padding_mode}")
    # The kernels address each tensor with 32-bit byte offsets (< 2 GiB).  Larger activations (40 folded frames x 128
    # channels at 256x512 = 2.7 GB) are run as batch chunks: images are independent in the forward pass and the data
    # gradient, and autograd sums the chunks' weight gradients.
    n = x.shape[0]
    k = _chunks_for_2gib(x.shape, w.shape, stride3, pad3)
    if k > 1:
        size = _cdiv(n, k)
        return torch.cat([_ConvFn.apply(xc, w, b, stride3, pad3, reflect, act, dgrad_channels, slope) for xc in x.split(size)], 0)
    return _ConvFn.apply(x, w, b, stride3, pad3, reflect, act, dgrad_channels, slope)


class _ConvReluTapFn(torch.autograd.Function):
    """y = relu(conv3x3(x; frozen w, b)) together with the L1 tap mean|y - t| it feeds (a perceptual-loss tap of the frozen
    VGG-19, losses.py:60-65): backward folds the L1 gradient, its sum with the next conv's data gradient and the ReLU mask
    into ONE element-wise pass (c2m_relu_tap_bwd) in front of the data gradient -- autograd ran three (9 tensor passes instead
    of 4 over the largest tensors of the step).  Same fp32 arithmetic, so values and gradients are bit-identical to
    conv(act='relu') + l1_mean."""

    @staticmethod
    def forward(ctx, x, w, b, t, pad3):
        y = _ConvFn.apply(x, w, b, (1, 1, 1), pad3, False, "relu", None, LRELU_SLOPE)       # (grad mode is off in here)
        t = _as(t, y.dtype)
        l = _L1MeanFn.apply(y, t, None)
        ctx.pl = _plan(_f(x), _f(w), (1, 1, 1), pad3, False, None)
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(w, y, t)
        ctx.set_materialize_grads(False)
        return y, l

    @staticmethod
    def backward(ctx, gy, gl):
        if gy is None and gl is None:
            return None, None, None, None, None
        w, y, t = ctx.saved_tensors
        if gl is None:
            gl = torch.zeros((), device=y.device, dtype=torch.float32)
        gl = _f(gl.reshape(1).float())
        gy = None if gy is None else _as(gy, y.dtype)
        if y.dtype == BF16 and _bwd_reads_only_nc8(ctx.pl, True, False):
            Nn, Cc = y.shape[0], y.shape[1]
            gn = torch.empty((Nn, _cdiv(Cc, 8)) + tuple(y.shape[2:]) + (8,), device=y.device, dtype=BF16)
            _lib.check(_lib.lib().c2m_grad_to_nc8(1, _p(y), _p(t), _p(gy), _p(gl), _p(gn), Nn, Cc, y.numel() // (Nn * Cc), y.numel(),
                                                  0, 0.0, _stream()), "grad_to_nc8 (relu_tap_bwd)")
            return _conv_dgrad(ctx.pl, w, _virtual_grad(y, gn), True, ctx.x_dtype), None, None, None, None
        g = torch.empty_like(y)
        _lib.check(_lib.lib().c2m_relu_tap_bwd(_p(y), _p(t), _p(gy), _p(gl), _p(g), y.numel(), _dt(y), _stream()), "relu_tap_bwd")
        if _NC8_LOG is not None:
            _NC8_LOG_TAG[0] = " (relu_tap_bwd)"
        return _conv_dgrad(ctx.pl, w, g, True, ctx.x_dtype), None, None, None, None


def conv_relu_tap(x, w, b, target, padding=1):
    """(y, mean|y - target|) for y = relu(conv(x, w, b)) with a FROZEN w / b (the VGG-19 of the perceptual loss) and a target
    without gradient: conv(act='relu') + l1_mean with a fused backward (see _ConvReluTapFn).  Falls back to the two ops where
    the fused form does not apply (weights that need gradients, batches that run as 2 GiB chunks)."""
    nd = x.dim() - 2
    pad3 = _pad3(padding, nd)
    if w.requires_grad or (b is not None and b.requires_grad) or target.requires_grad or not x.requires_grad or \
            not torch.is_grad_enabled() or _chunks_for_2gib(x.shape, w.shape, (1, 1, 1), pad3) > 1:
        y = conv(x, w, b, stride=1, padding=padding, padding_mode="zeros", act="relu")
        return y, l1_mean(y, target)
    return _ConvReluTapFn.apply(x, w, b, target, pad3)


class _ConvReluPoolFn(torch.autograd.Function):
    """maxpool2x2(relu(conv3x3(x; frozen w, b))) as one node (layers/vgg.py: the conv -> ReLU -> MaxPool2d runs of the frozen
    VGG-19 whose ReLU output nobody else reads): the pool backward applies the ReLU mask on the value it already holds
    (c2m_maxpool2x2_relu_bwd), so the activation-backward pass over the full-resolution tensor disappears.  Same arithmetic ->
    bit-identical to conv(act='relu') + maxpool2x2."""

    @staticmethod
    def forward(ctx, x, w, b, pad3):
        y = _ConvFn.apply(x, w, b, (1, 1, 1), pad3, False, "relu", None, LRELU_SLOPE)
        p = _MaxPool2Fn.apply(y)
        ctx.pl = _plan(_f(x), _f(w), (1, 1, 1), pad3, False, None)
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(w, y)
        return p

    @staticmethod
    def backward(ctx, gp):
        w, y = ctx.saved_tensors
        N, C, H, W = y.shape
        g = torch.empty_like(y)
        _lib.check(_lib.lib().c2m_maxpool2x2_relu_bwd(_p(y), _p(_as(gp, y.dtype)), _p(g), N * C, H, W, _dt(y), _stream()),
                   "maxpool_relu_bwd")
        return _conv_dgrad(ctx.pl, w, g, True, ctx.x_dtype), None, None, None


def conv_relu_pool(x, w, b, padding=1):
    """maxpool2x2(relu(conv(x, w, b))) for a FROZEN w / b, with the ReLU backward folded into the pool backward; the unfused
    ops where that does not apply (trainable weights, no gradient wanted, 2 GiB batch chunks, odd extents)."""
    nd = x.dim() - 2
    pad3 = _pad3(padding, nd)
    if nd != 2 or w.requires_grad or (b is not None and b.requires_grad) or not x.requires_grad or not torch.is_grad_enabled() or \
            _chunks_for_2gib(x.shape, w.shape, (1, 1, 1), pad3) > 1:
        return maxpool2x2(conv(x, w, b, stride=1, padding=padding, padding_mode="zeros", act="relu"))
    return _ConvReluPoolFn.apply(x, w, b, pad3)


def conv_transpose2d(x, w, b=None, stride=2, padding=1, act=None, slope=LRELU_SLOPE):
    """nn.ConvTranspose2d(x; w [Cin, Cout, kh, kw], stride, padding) (+ bias, activation) for frozen, no-grad use (FlowNet2's
    `deconv` / `upsampled_flow` layers, flownet2/networks/submodules.py:75-80): a transposed convolution IS the data gradient
    of the convolution with the same weight tensor, so it runs on the stride-parity-class data-gradient kernels (no
    zero-stuffed MACs).  Output size (H - 1) * stride - 2 * padding + k."""
    _dev(x, w, b)
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        raise RuntimeError("conv_transpose2d is forward-only (frozen flow network under no_grad)")
    x, w = _f(x), _f(w)
    N, Ci, H, W = x.shape
    Co, kh, kw = w.shape[1], w.shape[2], w.shape[3]
    s3, p3 = _triple(stride, 2), _pad3(padding, 2)
    Ho, Wo = (H - 1) * s3[1] - 2 * p3[1] + kh, (W - 1) * s3[2] - 2 * p3[2] + kw
    # the forward convolution whose data gradient this is: input [N, Co, Ho, Wo] (x of the plan), weight [Ci, Co, kh, kw]
    key = ((N, Co, Ho, Wo), tuple(w.shape), s3, p3, False, x.device.index, _conv_bf16, None)
    pl = _geom_cache.get(key)
    if pl is None:
        pl = _geom_cache[key] = _ConvPlan((N, Co, Ho, Wo), tuple(w.shape), s3, p3, False, x.device, _conv_bf16, None)
        if _conv_bf16:
            pl.fwd_geom[G.PRECISION] = pl.wg_geom[G.PRECISION] = 1
            for c in pl.classes:
                c["geom"][G.PRECISION] = 1
            for grp in (pl.cls_batch["groups"] if pl.cls_batch else ()):
                grp["geom"][G.PRECISION] = 1
    if pl.out_shape != (N, Ci, H, W):
        raise ValueError(f"conv_transpose2d: inconsistent geometry {tuple(x.shape)} vs {pl.out_shape}")
    with torch.no_grad():
        y = _conv_dgrad(pl, w, x, True, x.dtype)
        if b is not None or ACT[act]:
            y = _as(y, torch.float32)
            _lib.check(_lib.lib().c2m_bias_act(_p(y), _p(b), N, Co, Ho * Wo, ACT[act], float(slope), _stream()), "bias_act")
    return y


_MAX_TENSOR_BYTES = int(0.9 * 2 ** 31)      # 10 % margin: the Winograd data gradient rounds its padded domain up to whole tiles


def _chunks_for_2gib(xs, ws, stride, pad):
    """Number of batch chunks that keeps the input (incl. its padded data-gradient domain) and the output of one
    launch under 2 GiB; 1 for everything the bench configurations use at 128x256."""
    n, cin, cout = xs[0], xs[1], ws[0]
    sp_in = [d for d in xs[2:]]
    k3 = list(ws[2:])
    nd = len(sp_in)
    padded, out = 1, 1
    for d in range(nd):
        padded *= sp_in[d] + 2 * pad[3 - nd + d]
        out *= max((sp_in[d] + 2 * pad[3 - nd + d] - k3[d]) // stride[3 - nd + d] + 1, 1)
    per_image = 4 * max(cin * padded, cout * out)
    if n * per_image < _MAX_TENSOR_BYTES:
        return 1
    fit = (_MAX_TENSOR_BYTES - 1) // per_image
    if fit < 1:
        raise ValueError("tensor too large: one image exceeds the 32-bit byte offsets of the gather (< 2 GiB)")
    return _cdiv(n, fit)


# =============================================================================================== norm + act
class _NormActFn(torch.autograd.Function):
    """y = act(norm(x) * scale + shift); mode 0 instance / 1 batch statistics; gb = SPADE [N,2C,...] map or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, gb, running_mean, running_var, mode, act, eps, momentum, feeds=None, private_input=False):
        _dev(x, gamma, beta, gb)
        x = _f(x)
        ctx.gb_dtype = gb.dtype if gb is not None else None
        gb = _as(gb, x.dtype)                   # the SPADE map shares the activation type (bf16 on the bf16 data path)
        dt = _dt(x)
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        # the kernels index these by channel: sizes checked here (torch's norm layers raise on the same mistakes)
        for name, t, want in (("gamma", gamma, C), ("beta", beta, C), ("running_mean", running_mean, C), ("running_var", running_var, C),
                              ("SPADE map", gb, 2 * x.numel())):
            if t is not None and t.numel() != want:
                raise RuntimeError(f"norm: {name} has {t.numel()} elements, expected {want} for an input of shape {tuple(x.shape)}")
        L = _lib.lib()
        nstat = N * C if mode == 0 else C
        mean = torch.empty(nstat, device=x.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        ws = torch.empty(L.c2m_norm_workspace_floats(N, C, S), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        # bf16 data path: activations of conv-sized maps are ALSO written in the NC8 layout of the convolution they feed (and, in
        # backward, the gradient they hand to the convolution in front): conv_nc8.hip then needs no layout pass of its own
        ctx.nc8 = bool(_NC8 and _NC8_NORM and dt == 1 and x.dim() == 4 and S % 8 == 0 and C >= 16 and x.shape[3] >= 16 and S >= 512)
        # `feeds`: the result's only consumer is one convolution whose forward and weight gradient read NC8 -> NC8 is the ONLY output
        only = False
        if feeds is not None and dt == 1 and S % 8 == 0:
            only = _consumer_reads_only_nc8(x, feeds)
        yn = torch.empty((N, _cdiv(C, 8)) + tuple(x.shape[2:]) + (8,), device=x.device, dtype=BF16) if (ctx.nc8 or only) else None
        if yn is None:
            # statistics + apply in one call: instance-norm planes of <= 32768 elements run as ONE launch (norm_inst_fused_kernel)
            _lib.check(L.c2m_norm_fwd(_p(x), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(ws), _p(gamma), _p(beta), _p(gb),
                                      _p(y), N, C, S, mode, eps, momentum, ACT[act], LRELU_SLOPE, dt, _stream()), "norm_fwd")
        else:
            _lib.check(L.c2m_norm_stats(_p(x), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(ws), N, C, S, mode,
                                        eps, momentum, dt, _stream()), "norm_stats")
            _lib.check(L.c2m_norm_apply(_p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(gb), None if only else _p(y), _p(yn), N, C, S,
                                        mode, ACT[act], LRELU_SLOPE, dt, _stream()), "norm_apply")
        if only and _NC8_POISON:
            y.fill_(float("nan"))
        _NormActFn.last_only = only             # read by _norm_act right behind apply(): tags y as NC8-only
        ctx.cfg = (N, C, S, mode, act)
        # x is the output of a convolution whose backward reads its gradient in NC8 only, and this op is x's only consumer (the
        # caller's promise): dx is produced in NC8 form alone
        ctx.dx_nc8_only = bool(private_input and _NC8_GRAD and dt == 1 and S % 8 == 0 and getattr(x, "_c2m_bwd_nc8", False))
        ctx.save_for_backward(x, gamma, beta, gb, mean, invstd)
        if yn is None:
            return y, None
        ctx.mark_non_differentiable(yn)
        return y, yn

    @staticmethod
    def backward(ctx, gy, _gyn=None):
        x, gamma, beta, gb, mean, invstd = ctx.saved_tensors
        N, C, S, mode, act = ctx.cfg
        L = _lib.lib()
        gy = _as(gy, x.dtype)
        only = ctx.dx_nc8_only
        dx = torch.empty_like(x)
        dxn = torch.empty((N, _cdiv(C, 8)) + tuple(x.shape[2:]) + (8,), device=x.device, dtype=BF16) if (ctx.nc8 or only) else None
        ggb = torch.empty_like(gb) if gb is not None else None
        dgamma = torch.empty_like(gamma) if gamma is not None else None
        dbeta = torch.empty_like(beta) if gamma is not None else None
        ws = torch.empty(L.c2m_norm_workspace_floats(N, C, S), device=x.device, dtype=torch.float32)
        _lib.check(L.c2m_norm_bwd(_p(x), _p(gy), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(gb), _p(ggb), _p(dgamma),
                                  _p(dbeta), None if only else _p(dx), _p(dxn), _p(ws), N, C, S, mode, ACT[act], LRELU_SLOPE, _dt(x),
                                  _stream()),
                   "norm_bwd")
        if only and _NC8_POISON:
            dx.fill_(float("nan"))
        if dxn is not None and only:
            _mark_nc8_only(dx, dxn)               # autograd hands this very tensor object to the convolution's backward (see _to_nc8)
        elif dxn is not None:
            dx._c2m_nc8 = (dx._version, dxn)
        if ggb is not None and ggb.dtype != ctx.gb_dtype:
            ggb = ggb.to(ctx.gb_dtype)
        return dx, dgamma, dbeta, ggb, None, None, None, None, None, None, None, None


def _norm_act(*args):
    y, yn = _NormActFn.apply(*args)
    if yn is not None and _NormActFn.last_only:
        _mark_nc8_only(y, yn)                     # y's NCHW storage was never written
    elif yn is not None:
        y._c2m_nc8 = (y._version, yn)             # the NC8 form travels with the tensor object (ops._to_nc8 picks it up)
    return y


def batch_norm_act(x, gamma, beta, running_mean, running_var, act=None, eps=1e-5, momentum=0.1, feeds=None, private_input=False):
    """private_input: x is a convolution's output that NOTHING else reads (`y = conv(..); y = norm(y)` inside a block) -- where that
    convolution's backward runs on NC8 kernels, dx is handed back in NC8 form only (no NCHW write, no layout pass)."""
    return _norm_act(x, gamma, beta, None, running_mean, running_var, 1, act, eps, momentum, feeds, private_input)


def instance_norm_act(x, gamma=None, beta=None, act=None, eps=1e-5, feeds=None, private_input=False):
    return _norm_act(x, gamma, beta, None, None, None, 0, act, eps, 0.1, feeds, private_input)


def spade_norm_act(x, gamma_beta, act=None, eps=1e-5, feeds=None, private_input=False):
    """InstanceNorm(affine=False)(x) * (1 + gamma) + beta with [gamma, beta] = gamma_beta.chunk(2, 1), then act.
    feeds (all three ops): `conv_consumer(...)` of the one convolution that reads the result, see there."""
    return _norm_act(x, None, None, gamma_beta, None, None, 0, act, eps, 0.1, feeds, private_input)


# =============================================================================================== warping / resampling
class _FlowWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow, occ):
        _dev(img, flow, occ)
        ctx.flow_dtype = flow.dtype
        img, flow = _f(img), _as(flow, torch.float32)        # coordinates are never rounded to bf16
        occ = _as(occ, torch.float32)
        N, C, H, W = img.shape
        assert flow.shape == (N, 2, H, W), f"flow {tuple(flow.shape)} vs image {tuple(img.shape)}"
        assert occ is None or occ.shape == (N, 1, H, W)
        out = torch.empty_like(img)
        _lib.check(_lib.lib().c2m_flow_warp_fwd(_p(img), _p(flow), _p(occ), _p(out), N, C, H, W, _dt(img), _stream()),
                   "flow_warp")
        ctx.save_for_backward(img, flow, occ)
        return out

    @staticmethod
    def backward(ctx, gout):
        img, flow, occ = ctx.saved_tensors
        N, C, H, W = img.shape
        L = _lib.lib()
        gimg = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if gimg is not None or gflow is not None:
            ws = torch.empty(L.c2m_flow_warp_bwd_workspace_bytes(N, C, H, W, int(gimg is not None), int(gflow is not None)),
                             device=img.device, dtype=torch.uint8)
            _lib.check(L.c2m_flow_warp_bwd(_p(img), _p(flow), _p(occ), _p(_as(gout, img.dtype)), _p(gimg), _p(gflow), N, C, H,
                                           W, _p(ws), _dt(img), _stream()), "flow_warp_bwd")
        if gflow is not None and gflow.dtype != ctx.flow_dtype:
            gflow = gflow.to(ctx.flow_dtype)
        return gimg, gflow, None


def flow_warp(img, flow, occ=None):
    """utils.resample(img, flow) [* occ]: backward warp with pixel-unit flow (reference coordinate quirk included)."""
    return _FlowWarpFn.apply(img, flow, occ)


def resample2d(img, flow):
    """FlowNet2 Resample2d (kernel_size 1, bilinear): img [N,C,H,W] sampled at (x + flow[:,0], y + flow[:,1]), taps clamped to
    the border (third_party/resample2d/src/resample2d_kernel.cu:16-75).  Forward only."""
    _dev(img, flow)
    img, flow = _as(img.detach(), torch.float32), _as(flow.detach(), torch.float32)     # fp32-only kernel (bf16 conv outputs are cast)
    N, C, H, W = img.shape
    assert flow.shape == (N, 2, H, W)
    out = torch.empty_like(img)
    _lib.check(_lib.lib().c2m_resample2d_fwd(_p(img), _p(flow), _p(out), N, C, H, W, _stream()), "resample2d")
    return out


def channelnorm(x):
    """FlowNet2 ChannelNorm: sqrt(sum_c x^2) -> [N,1,H,W] (third_party/channelnorm/src/channelnorm_kernel.cu:19-62)."""
    _dev(x)
    x = _as(x.detach(), torch.float32)                   # fp32-only kernel
    N, C, H, W = x.shape
    out = torch.empty(N, 1, H, W, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().c2m_channelnorm_fwd(_p(x), _p(out), N, C, H, W, _stream()), "channelnorm")
    return out


def correlation(a, b, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    """FlowNetC cost volume (third_party/correlation/src/correlation_cuda_kernel.cu:47-147; defaults = flownet_c.py:44-46):
    [N,C,H,W] x [N,C,H,W] -> [N, (2*(max_displacement//stride2)+1)^2, oH, oW]."""
    _dev(a, b)
    a, b = _as(a.detach(), torch.float32), _as(b.detach(), torch.float32)      # fp32-only kernel: a bf16 pointer would be read as floats
    assert a.shape == b.shape
    N, C, H, W = a.shape
    L = _lib.lib()
    D = 2 * (max_displacement // stride2) + 1
    oH = L.c2m_correlation_out_size(H, pad_size, kernel_size, max_displacement, stride1)
    oW = L.c2m_correlation_out_size(W, pad_size, kernel_size, max_displacement, stride1)
    out = torch.empty(N, D * D, oH, oW, device=a.device, dtype=torch.float32)
    _lib.check(L.c2m_correlation_fwd(_p(a), _p(b), _p(out), N, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                                     stride2, _stream()), "correlation")
    return out


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, feeds=None):
        _dev(x)
        x = _f(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
        ctx.shape = (N, C, H, W)
        if feeds is not None and x.dtype == BF16 and (H * W) % 8 == 0:
            # the result's only reader is one convolution whose forward and weight gradient read NC8: NC8 -> NC8, no NCHW result
            if _consumer_reads_only_nc8(y, feeds):
                xn = _to_nc8(x)
                yn = torch.empty((N, _cdiv(C, 8), 2 * H, 2 * W, 8), device=x.device, dtype=BF16)
                _lib.check(_lib.lib().c2m_upsample2x_nc8(_p(xn), _p(yn), N * _cdiv(C, 8), H, W, _stream()), "upsample2x_nc8")
                if _NC8_POISON:
                    y.fill_(float("nan"))
                _Upsample2xFn.last_form = yn      # read by upsample2x() right behind apply(): tags y as NC8-only
                return y
        _Upsample2xFn.last_form = None
        _lib.check(_lib.lib().c2m_upsample2x_fwd(_p(x), _p(y), N * C, H, W, _dt(x), _stream()), "upsample2x")
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = _f(gy)
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=gy.dtype)
        _lib.check(_lib.lib().c2m_upsample2x_bwd(_p(gy), _p(gx), N * C, H, W, _dt(gy), _stream()), "upsample2x_bwd")
        return gx, None


def upsample2x(x, feeds=None):
    """nn.Upsample(scale_factor=2, mode='bilinear').  feeds: `conv_consumer(...)` of the ONE convolution that reads the result (the up
    block): where that layer reads NC8 only, the up-sampled map -- the largest tensors of the decoder -- is produced in NC8 alone."""
    y = _Upsample2xFn.apply(x, feeds)
    if _Upsample2xFn.last_form is not None:
        _mark_nc8_only(y, _Upsample2xFn.last_form)
        _Upsample2xFn.last_form = None
    return y


def resize_bilinear(x, size, align_corners=False):
    """F.interpolate(x, size, mode='bilinear') for no-grad tensors (sparse flow / occlusion pyramids)."""
    _dev(x)
    if x.requires_grad:
        raise RuntimeError("resize_bilinear has no backward; use upsample2x for the differentiable x2 case")
    x = _f(x)
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
    _lib.check(_lib.lib().c2m_resize_bilinear(_p(x), _p(y), N * C, H, W, Ho, Wo, 1 if align_corners else 0, 0.0,
                                              _dt(x), _stream()), "resize_bilinear")
    return y


class _MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _f(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, H // 2, W // 2, device=x.device, dtype=x.dtype)
        _lib.check(_lib.lib().c2m_maxpool2x2_fwd(_p(x), _p(y), N * C, H, W, _dt(x), _stream()), "maxpool")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().c2m_maxpool2x2_bwd(_p(x), _p(_as(gy, x.dtype)), _p(gx), N * C, H, W, _dt(x), _stream()),
                   "maxpool_bwd")
        return gx


def maxpool2x2(x):
    return _MaxPool2Fn.apply(x)


class _RoiAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, boxes, ph, pw, scale):
        _dev(feat, boxes)
        ctx.feat_dtype = feat.dtype
        feat, boxes = _as(feat, torch.float32), _as(boxes, torch.float32)      # a [B,64,32,64] map: fp32 kernel, cast once
        N, C, H, W = feat.shape
        K = boxes.shape[0]
        out = torch.empty(K, C, ph, pw, device=feat.device, dtype=torch.float32)
        _lib.check(_lib.lib().c2m_roi_align_fwd(_p(feat), _p(boxes), _p(out), K, C, H, W, ph, pw, scale, _stream()),
                   "roi_align")
        ctx.cfg = (N, C, H, W, K, ph, pw, scale)
        ctx.save_for_backward(boxes)
        return out

    @staticmethod
    def backward(ctx, gout):
        (boxes,) = ctx.saved_tensors
        N, C, H, W, K, ph, pw, scale = ctx.cfg
        gfeat = torch.empty(N, C, H, W, device=gout.device, dtype=torch.float32)
        _lib.check(_lib.lib().c2m_roi_align_bwd(_p(boxes), _p(_as(gout, torch.float32)), _p(gfeat), N, K, C, H, W, ph, pw,
                                                scale, _stream()), "roi_align_bwd")
        return gfeat if gfeat.dtype == ctx.feat_dtype else gfeat.to(ctx.feat_dtype), None, None, None, None


def roi_align(feat, boxes, output_size, spatial_scale=1.0):
    """torchvision.ops.roi_align(feat, boxes[K,5], output_size, spatial_scale) with aligned=False, sampling_ratio=-1."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    return _RoiAlignFn.apply(feat, boxes.to(torch.float32), int(ph), int(pw), float(spatial_scale))


# =============================================================================================== object GNN
class _GatDenseFn(torch.autograd.Function):
    """Attention of a GATv2 layer on a dense graph of <= 64 nodes (csrc/gnn.hip): one launch forward, two backward."""

    @staticmethod
    def forward(ctx, xl, xr, att, A, slope):
        _dev(xl, xr, att, A)
        xl, xr, att, A = (_as(t, torch.float32) for t in (xl, xr, att, A))
        N, H, C = xl.shape
        assert xr.shape == (N, H, C) and att.shape == (H, C) and A.shape == (N, N)
        out = torch.empty(N, C, device=xl.device, dtype=torch.float32)
        alpha = torch.empty(N, N, H, device=xl.device, dtype=torch.float32)
        _lib.check(_lib.lib().c2m_gat_dense_fwd(_p(xl), _p(xr), _p(att), _p(A), _p(out), _p(alpha), N, H, C, float(slope), _stream()),
                   "gat_dense_fwd")
        ctx.slope = float(slope)
        ctx.save_for_backward(xl, xr, att, alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        xl, xr, att, alpha = ctx.saved_tensors
        N, H, C = xl.shape
        L = _lib.lib()
        dxl, dxr, datt = torch.empty_like(xl), torch.empty_like(xr), torch.empty_like(att)
        ws = torch.empty(L.c2m_gat_dense_bwd_workspace_floats(N, H, C), device=xl.device, dtype=torch.float32)
        _lib.check(L.c2m_gat_dense_bwd(_p(xl), _p(xr), _p(att), _p(alpha), _p(_as(g, torch.float32)), _p(dxl), _p(dxr), _p(datt), _p(ws),
                                       N, H, C, ctx.slope, _stream()), "gat_dense_bwd")
        return dxl, dxr, datt, None, None


GAT_DENSE_MAX_NODES, GAT_DENSE_MAX_CHANNELS = 64, 1024
_GAT_FUSED = os.environ.get("C2M_GAT_FUSED", "1") != "0"


def gatv2_dense_ok(xl, N, C):
    return _GAT_FUSED and xl.is_cuda and xl.dtype == torch.float32 and N <= GAT_DENSE_MAX_NODES and C <= GAT_DENSE_MAX_CHANNELS


def gatv2_dense(xl, xr, att, A, negative_slope=0.2):
    """mean over heads of softmax_j(att . lrelu(xl[j] + xr[i]); weights A[i, j]) applied to xl -- the message passing of
    GATv2Conv(concat=False, add_self_loops=False) for all ordered node pairs (thirdparty.GATv2Conv adds the bias).  xl, xr [N, H, C],
    att [H, C], A [N, N] edge multiplicities; returns [N, C]."""
    return _GatDenseFn.apply(xl, xr, att, A, negative_slope)


# =============================================================================================== index / mask path
def sparse_raster(instance, obj_id, obj_batch, thetas):
    """instance [B,H,W] float ids, obj_id/obj_batch [K], thetas [K,T,6] -> (bw [B,2,T,H,W], fw, bin [B,1,T,H,W])."""
    _dev(instance, thetas)
    instance, thetas = _as(instance, torch.float32), _as(thetas.detach(), torch.float32)
    B, H, W = instance.shape
    K, T = thetas.shape[0], thetas.shape[1]
    oid = obj_id.to(device=instance.device, dtype=torch.int32).contiguous()
    ob = obj_batch.to(device=instance.device, dtype=torch.int32).contiguous()
    bw = torch.empty(B, 2, T, H, W, device=instance.device, dtype=torch.float32)
    fw = torch.empty_like(bw)
    binm = torch.empty(B, 1, T, H, W, device=instance.device, dtype=torch.float32)
    _lib.check(_lib.lib().c2m_sparse_raster(_p(instance), _p(oid), _p(ob), _p(thetas), _p(bw), _p(fw), _p(binm), B, K, T,
                                            H, W, _stream()), "sparse_raster")
    return bw, fw, binm


def occlusion_splat(flow, want_map=True, want_clip=False):
    """get_occlusion_map for [B,2,H,W] or, frame-batched, [B,2,T,H,W] flows -> ([B,1,(T,)H,W] map, clip_mask) ."""
    _dev(flow)
    flow = _as(flow.detach(), torch.float32)
    five = flow.dim() == 5
    if five:
        B, _, T, H, W = flow.shape
        sb, sc, st = 2 * T * H * W, T * H * W, H * W
        oshape = (B, 1, T, H, W)
    else:
        B, _, H, W = flow.shape
        T, sb, sc, st = 1, 2 * H * W, H * W, 0
        oshape = (B, 1, H, W)
    L = _lib.lib()
    ws = torch.empty(L.c2m_occlusion_splat_workspace_bytes(B * T, H, W), device=flow.device, dtype=torch.uint8)
    occ = torch.empty(oshape, device=flow.device, dtype=torch.float32) if want_map else None
    clip = torch.empty(oshape, device=flow.device, dtype=torch.float32) if want_clip else None
    _lib.check(L.c2m_occlusion_splat(_p(flow), sb, sc, st, B, T, H, W, _p(occ), _p(clip), _p(ws), _stream()),
               "occlusion_splat")
    return occ, clip


# =============================================================================================== losses
class _L1MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask):
        _dev(a, b, mask)
        ctx.b_dtype = b.dtype
        a = _f(a)
        b = _as(b, a.dtype)
        assert a.shape == b.shape
        C, inner = 1, 1
        if mask is not None:
            mask = _as(mask, torch.float32)
            assert mask.shape[0] == a.shape[0] and mask.shape[1] == 1 and mask.shape[2:] == a.shape[2:]
            C, inner = a.shape[1], a[0, 0].numel()
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = torch.empty(1024, device=a.device, dtype=torch.float64)
        _lib.check(_lib.lib().c2m_l1_mean_fwd(_p(a), _p(b), _p(mask), _p(out), a.numel(), C, inner, _p(ws), _dt(a), _stream()),
                   "l1_mean")
        ctx.cfg = (C, inner)
        ctx.save_for_backward(a, b, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, mask = ctx.saved_tensors
        C, inner = ctx.cfg
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        g = _f(g.reshape(1).float())
        _lib.check(_lib.lib().c2m_l1_mean_bwd(_p(a), _p(b), _p(mask), _p(g), _p(ga), _p(gb), a.numel(), C, inner,
                                              _dt(a), _stream()), "l1_mean_bwd")
        if gb is not None and gb.dtype != ctx.b_dtype:
            gb = gb.to(ctx.b_dtype)
        return ga, gb, None


def l1_mean(a, b, mask=None):
    """F.l1_loss(a*mask, b*mask) with mask [B,1,...] broadcast over channels (L1MaskedLoss), or plain L1 mean."""
    return _L1MeanFn.apply(a, b, mask)


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        _dev(x, y)
        ctx.x_dtype = x.dtype
        x, y = _as(x, torch.float32), _as(y, torch.float32)
        N, C, H, W = x.shape
        out = torch.empty((), device=x.device, dtype=torch.float32)
        ws = torch.empty(1024, device=x.device, dtype=torch.float64)
        _lib.check(_lib.lib().c2m_ssim_fwd(_p(x), _p(y), _p(out), N * C, H, W, _p(ws), _stream()), "ssim")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        coef = torch.empty(N * C * (H - 2) * (W - 2) * 3, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().c2m_ssim_bwd(_p(x), _p(y), _p(_f(g.reshape(1).float())), _p(gx), _p(coef), N * C, H, W,
                                           _stream()), "ssim_bwd")
        return gx if gx.dtype == ctx.x_dtype else gx.to(ctx.x_dtype), None


def ssim_loss(x, y):
    """SSIMLoss.ssim on [N,C,H,W] (x = generated, differentiable; y = target)."""
    return _SsimFn.apply(x, y)


def norm_apply_eval(x, mean, invstd, gamma, beta, act=None):
    """Inference-mode BatchNorm (+act): per-channel affine with given statistics. No autograd."""
    _dev(x, mean, invstd, gamma, beta)
    x = _f(x)
    N, C = x.shape[0], x.shape[1]
    S = x.numel() // (N * C)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().c2m_norm_apply(_p(x), _p(_f(mean)), _p(_f(invstd)), _p(gamma), _p(beta), None, _p(y), None, N, C, S, 1,
                                         ACT[act], LRELU_SLOPE, _dt(x), _stream()), "norm_apply(eval)")
    return y
