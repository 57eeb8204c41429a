"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the sample-building arithmetic of the reference's dataset class
(src/datasets/cityscapes.py), from decoded arrays to the tensors of the batch dict.
PINNED (round 3): tests/golden/data_dataset_prep.npz holds what the LIVE reference's own functions
(cityscapes.py:20-70 replace_index_and_read_frame / read_video, :195-199 load_tracking_mask, :208-231 load_instance /
load_optical_flow, :234-265 load_optical_flow_occlusion_mask / clip_mask) built from PNG / .flo files that
oracle/capture_golden.py::capture_dataset wrote from the stored arrays (PIL decodes them; the one stand-in is torchvision's
ToTensor -- an absent, un-vendored third-party dependency whose published behaviour is restated: uint8 HWC -> CHW float32
`.div(255)`, other dtypes only transposed); tests/test_data_graph.py holds every function below to it bit for bit.
The scene-graph and .flo functions at the end of the file are pinned by tests/golden/data_scene_graph.npz (live reference on
committed track files).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
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


# ---------------------------------------------------------------------------------------------- scene graph
# Pinned against the live reference by tests/golden/data_scene_graph.npz (oracle/capture_golden.py::capture_data).
def scene_info(tracks, size, t_in, num_frames, lambda_traj=1):
    """cityscapes.py:79-193 as scalar Python loops (one instance = its list of "x,y,w,h,...,id" lines).
    Returns a dict of float32 / int64 numpy arrays with the field names of the reference's Data object."""
    import numpy as np
    from itertools import permutations
    H, W = size
    out = {k: [] for k in ("x", "roi", "roi_pad", "src_ids", "tgt_ids", "tgt_roi", "bary", "disp", "theta")}
    for lines in tracks:
        rows = [ln.split(",") for ln in lines[:num_frames]]
        feat, roi_s, pad_s, ids_s, ids_t, roi_t, bary_t, disp_t, theta_t = [], [], [], [], [], [], [], [], []
        last_bary = last_size = None
        for idx, c in enumerate(rows):
            bx, by, bw, bh = (float(v) for v in c[:4])
            x_l, x_r = bx / 2048 * W, (bx + bw) / 2048 * W                        # :113-116
            y_t, y_b = by / 1024 * H, (by + bh) / 1024 * H                        # :117-120
            x_c = (x_l + x_r) / 2
            if idx >= t_in and lambda_traj > 1:                                   # :128-141
                s = rows[t_in - 1]
                xs = (float(s[0]) / 2048 * W + (float(s[0]) + float(s[2])) / 2048 * W) / 2
                d = (x_c - xs) * lambda_traj
                x_c, x_l, x_r = xs + d, x_l + d, x_r + d
            pad = [max(x_l - 15, 0), min(x_r + 15, W), max(y_t - 10, 0), min(y_b + 10, H)]
            bary = np.array([(y_t + y_b) / 2 / H * 2 - 1, x_c / W * 2 - 1])       # :142
            sz = np.array([bh / 1024, bw / 2048])                                 # :121
            if idx < t_in:                                                        # :143-151
                feat.append([bary[0], bary[1], sz[0], sz[1]] + list(np.eye(19)[int(c[-1]) // 1000]))
                ids_s.append(int(c[-1])); roi_s.append([x_l, x_r, y_t, y_b]); pad_s.append(pad)
                last_bary, last_size = bary, sz
            else:                                                                 # :152-160
                d = last_bary - bary
                sc = last_size / sz
                ids_t.append(int(c[-1])); roi_t.append([x_l, x_r, y_t, y_b]); bary_t.append(bary); disp_t.append(d)
                theta_t.append([sc[1], 0, d[1], 0, sc[0], d[0]])
        for k, v in zip(out, (feat, roi_s, pad_s, ids_s, ids_t, roi_t, bary_t, disp_t, theta_t)):
            out[k].append(v)
    n = len(tracks)
    res = {k: np.asarray(v, dtype=np.int64 if k.endswith("ids") else np.float32) for k, v in out.items()}
    res["edge_index"] = np.asarray(list(permutations(range(n), 2)) or [[0, 0]], dtype=np.int64).T
    res["tracking_ids"] = np.concatenate([res["src_ids"], res["tgt_ids"]], 1).T
    return res


def read_flo(path):
    """utils.py:324-343."""
    import numpy as np
    with open(path, "rb") as f:
        if np.fromfile(f, np.float32, count=1) != 202021.25:
            return None
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        return np.resize(np.fromfile(f, np.float32, count=2 * w * h), (h, w, 2))
