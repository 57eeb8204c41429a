"""Device-side batch assembly: the input-pipeline stage just upstream of the hot path.

Reference: src/datasets/cityscapes.py builds every sample on the CPU -- ToTensor of the frames (:30-33), the 20-channel
one-hot split of the label-id map (:35-41), instance ids (:43-52), occlusion PNG -> clip_mask (:212-216, 262-265), .flo
HWC -> CHW (:219-231) -- and src/train.py:23-38 collates.  Here the decoded arrays (what PIL / np.fromfile return, after
the resize) are uploaded as they are (uint8 / int32 / float32) and expanded on the device by three small kernels; the
result is the batch dict `GeneratorFullModel.forward` consumes.  File decoding / PIL resizing stay on the host."""
import torch

from . import _lib
from .ops import _p, _stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("c2m_amd.data needs tensors on a HIP device (no CPU fallback by design)")


def _u8(t, name):
    if t.dtype != torch.uint8:
        raise TypeError(f"{name} must be uint8 (decoded image data)")
    _dev(t)
    return t.contiguous()


def prep_video(frames_u8):
    """[B,T,H,W,3] uint8 -> video [B,3,T,H,W] float32 in [0,1] (ToTensor + stack over time, cityscapes.py:30-33,59-61)."""
    frames_u8 = _u8(frames_u8, "frames")
    B, T, H, W, C = frames_u8.shape
    if C != 3:
        raise ValueError("frames must be [B,T,H,W,3]")
    out = torch.empty(B, 3, T, H, W, device=frames_u8.device, dtype=torch.float32)
    _lib.check(_lib.lib().c2m_prep_video(_p(frames_u8), _p(out), B, T, H, W, _stream()), "prep_video")
    return out


def prep_seg_onehot(labels_u8):
    """[B,T,H,W] uint8 label ids -> (bg_mask [B,11,T,H,W], fg_mask [B,9,T,H,W]) (cityscapes.py:35-41,62-70)."""
    labels_u8 = _u8(labels_u8, "labels")
    B, T, H, W = labels_u8.shape
    bg = torch.empty(B, 11, T, H, W, device=labels_u8.device, dtype=torch.float32)
    fg = torch.empty(B, 9, T, H, W, device=labels_u8.device, dtype=torch.float32)
    _lib.check(_lib.lib().c2m_prep_seg_onehot(_p(labels_u8), _p(bg), _p(fg), B, T, H, W, _stream()), "prep_seg_onehot")
    return bg, fg


def prep_flow_occ(occ_u8, flow_hwc):
    """occlusion PNGs [B,T,H,W] uint8 and .flo arrays [B,T,H,W,2] float32 -> (target_bw_occ [B,1,T,H,W],
    target_bw_of [B,2,T,H,W]) (cityscapes.py:212-231,254-265).  Either input may be None."""
    ref = occ_u8 if occ_u8 is not None else flow_hwc
    _dev(ref)
    B, T, H, W = ref.shape[:4]
    occ = flow = None
    if occ_u8 is not None:
        occ_u8 = _u8(occ_u8, "occlusion")
        occ = torch.empty(B, 1, T, H, W, device=ref.device, dtype=torch.float32)
    if flow_hwc is not None:
        _dev(flow_hwc)
        if flow_hwc.dtype != torch.float32 or tuple(flow_hwc.shape) != (B, T, H, W, 2):
            raise ValueError("flow must be float32 [B,T,H,W,2]")
        flow_hwc = flow_hwc.contiguous()
        flow = torch.empty(B, 2, T, H, W, device=ref.device, dtype=torch.float32)
    _lib.check(_lib.lib().c2m_prep_flow_occ(_p(occ_u8), _p(flow_hwc), _p(occ), _p(flow), B, T, H, W, _stream()),
               "prep_flow_occ")
    return occ, flow


def assemble_batch(frames_u8, labels_u8, instance_i32, target_occ_u8, target_flow_hwc, tracking_gnn,
                   input_occ_u8=None, input_flow_hwc=None):
    """The batch dict of model.py:124 from decoded arrays already on the device.  `instance_i32` [B,T,H,W] int32."""
    bg, fg = prep_seg_onehot(labels_u8)
    occ, flow = prep_flow_occ(target_occ_u8, target_flow_hwc)
    batch = dict(video=prep_video(frames_u8), bg_mask=bg, fg_mask=fg,
                 instance_mask=instance_i32.to(torch.int32).unsqueeze(1).contiguous(), tracking_gnn=tracking_gnn,
                 target_bw_of=flow, target_bw_occ=occ, input_of=None, input_occ=None)
    if input_flow_hwc is not None:
        batch["input_occ"], batch["input_of"] = prep_flow_occ(input_occ_u8, input_flow_hwc)
    return batch
