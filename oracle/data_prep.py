"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the sample-building arithmetic of the reference's dataset class
(src/datasets/cityscapes.py), from decoded arrays to the tensors of the batch dict.  torchvision's ToTensor is an absent
third-party dependency (unpinned); its published behaviour is restated: uint8 HWC ndarray -> CHW float32
`.div(255)`; non-uint8 arrays are only transposed.  "parity unpinned" for this file: the reference has no fixtures for its
dataset code and needs image files to run.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import torch


def to_tensor(arr):
    """torchvision.transforms.ToTensor on an ndarray / tensor [H,W,C] or [H,W]."""
    t = torch.as_tensor(arr)
    if t.dim() == 2:
        t = t[:, :, None]
    t = t.permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def read_video(frames_u8):
    """cityscapes.py:59-61: stack ToTensor(frame) along dim 1.  frames [T,H,W,3] uint8 -> [3,T,H,W]."""
    return torch.stack([to_tensor(f) for f in frames_u8], dim=1)


def read_seg_masks(labels_u8):
    """cityscapes.py:35-41,62-70: seg = ToTensor(frame) * 255; fg = cat(seg == i, i in 11..19); bg = cat(seg == i, 0..10)."""
    fg, bg = [], []
    for f in labels_u8:
        seg = to_tensor(f) * 255
        fg.append(torch.cat([seg == i for i in range(11, 20)], 0).contiguous().type(torch.FloatTensor))
        bg.append(torch.cat([seg == i for i in range(0, 11)], 0).contiguous().type(torch.FloatTensor))
    return torch.stack(bg, dim=1), torch.stack(fg, dim=1)


def clip_mask(mask):
    """cityscapes.py:262-265."""
    return torch.where(mask > 0.5, torch.ones_like(mask), torch.zeros_like(mask))


def load_flow_occ(occ_u8, flow_hwc):
    """cityscapes.py:212-231,254-255: occlusion = clip_mask(stack(ToTensor(png))); flow = stack(flo.permute(2,0,1))."""
    occ = clip_mask(torch.stack([to_tensor(o) for o in occ_u8], dim=1))
    flow = torch.stack([torch.as_tensor(f).permute(2, 0, 1) for f in flow_hwc], dim=1)
    return occ, flow
