"""Host side of the input pipeline: tracking files -> per-scene object graph -> collated graph batch,
Middlebury .flo parsing and the tracking mask, without torch_geometric.

Reference: src/datasets/cityscapes.py:79-199 (`load_scene_info`, `load_tracking_mask`), src/train.py:23-38 (collate via
`Batch.from_data_list`), src/utils/utils.py:324-343 (`read_flow`).  The model only reads attributes of `tracking_gnn`
(SURVEY §8b), so plain attribute bags stand in for PyG's Data / Batch.  All geometry is done in float64 in the reference's
operation order and rounded to float32 once (torch.FloatTensor(list) does the same), so the tensors are bit-identical.
"""
import glob
from itertools import permutations

import numpy as np
import torch

from .synthetic import GraphBatch

_NODE_FIELDS = ("x", "y", "source_frames_nodes_roi", "source_frames_nodes_roi_padded", "target_frames_nodes_roi",
                "source_frames_nodes_instance_ids", "target_frames_nodes_instance_ids", "targets_barycenter",
                "targets_displacement", "targets_theta")


class GraphData:
    """One scene's graph (stand-in for torch_geometric.data.Data; attribute access only)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def num_nodes(self):
        return int(self.x.shape[0])


def parse_tracks(tracks, num_frames):
    """tracks: one entry per object instance = its text lines "x,y,w,h,...,id" (one line per frame; the first
    num_frames are used, cityscapes.py:107).  Returns boxes [N,T,4] float64 (x, y, w, h in 2048x1024 pixels) and the
    last column ids [N,T] int64."""
    boxes, ids = [], []
    for lines in tracks:
        lines = [ln for ln in lines][:num_frames]
        if len(lines) < num_frames:
            raise ValueError(f"a track has {len(lines)} frames, {num_frames} needed")
        cols = [ln.strip().split(",") for ln in lines]
        boxes.append([[float(c[0]), float(c[1]), float(c[2]), float(c[3])] for c in cols])
        ids.append([int(c[-1]) for c in cols])
    return np.asarray(boxes, dtype=np.float64).reshape(len(tracks), num_frames, 4), \
        np.asarray(ids, dtype=np.int64).reshape(len(tracks), num_frames)


def scene_graph(tracks, size, num_input_frames, num_frames, lambda_traj=1):
    """cityscapes.py:79-193 for one scene.  size = (H, W).  Returns (tracking_ids [T,N] int64, GraphData).

    Node features: [cy, cx] in [-1,1], [h, w] as image fractions, one-hot(19) of id // 1000 (ids >= 19000 raise
    IndexError like np.eye(19)[...] does).  targets_theta[n, t] = [sx, 0, dx, 0, sy, dy] relative to the LAST input
    frame.  lambda_traj > 1 stretches the horizontal displacement of the target frames (:126-141)."""
    H, W = size
    t_in = num_input_frames
    if not tracks:
        raise ValueError("a scene needs at least one tracked instance (the reference cannot collate an empty graph)")
    box, ids = parse_tracks(tracks, num_frames)
    N, T = ids.shape
    if not 1 <= t_in < T:
        raise ValueError("need 1 <= num_input_frames < num_frames")
    x_l = box[..., 0] / 2048 * W
    x_r = (box[..., 0] + box[..., 2]) / 2048 * W
    y_t = box[..., 1] / 1024 * H
    y_b = (box[..., 1] + box[..., 3]) / 1024 * H
    x_c = (x_l + x_r) / 2
    if lambda_traj > 1:
        start = x_c[:, t_in - 1:t_in]
        disp = (x_c[:, t_in:] - start) * lambda_traj
        x_c = np.concatenate([x_c[:, :t_in], start + disp], 1)
        x_l = np.concatenate([x_l[:, :t_in], x_l[:, t_in:] + disp], 1)
        x_r = np.concatenate([x_r[:, :t_in], x_r[:, t_in:] + disp], 1)
    y_c = (y_t + y_b) / 2
    roi = np.stack([x_l, x_r, y_t, y_b], -1)                               # stored [x_l, x_r, y_t, y_b] (App. A.8)
    roi_pad = np.stack([np.maximum(x_l - 15, 0), np.minimum(x_r + 15, W), np.maximum(y_t - 10, 0),
                        np.minimum(y_b + 10, H)], -1)
    bary = np.stack([y_c / H * 2 - 1, x_c / W * 2 - 1], -1)                # (y, x)
    bsize = np.stack([box[..., 3] / 1024, box[..., 2] / 2048], -1)        # (y, x)
    cls = ids[:, :t_in] // 1000
    if cls.min() < -19 or cls.max() >= 19:
        raise IndexError(f"index {int(cls.max())} is out of bounds for axis 0 with size 19")
    feats = np.concatenate([bary[:, :t_in], bsize[:, :t_in], np.eye(19)[cls]], -1)
    disp_t = bary[:, t_in - 1:t_in] - bary[:, t_in:]
    scale_t = bsize[:, t_in - 1:t_in] / bsize[:, t_in:]
    zeros = np.zeros_like(disp_t[..., 0])
    theta = np.stack([scale_t[..., 1], zeros, disp_t[..., 1], zeros, scale_t[..., 0], disp_t[..., 0]], -1)
    edges = list(permutations(range(N), 2)) or [[0, 0]]
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    src_ids, tgt_ids = torch.from_numpy(ids[:, :t_in].copy()), torch.from_numpy(ids[:, t_in:].copy())
    data = GraphData(x=f32(feats), y=f32(bary[:, t_in:]), num_real_nodes=torch.IntTensor([N]),
                     source_frames_nodes_roi=f32(roi[:, :t_in]), source_frames_nodes_roi_padded=f32(roi_pad[:, :t_in]),
                     target_frames_nodes_roi=f32(roi[:, t_in:]), source_frames_nodes_instance_ids=src_ids,
                     target_frames_nodes_instance_ids=tgt_ids, targets_barycenter=f32(bary[:, t_in:]),
                     targets_displacement=f32(disp_t), targets_theta=f32(theta),
                     edge_index=torch.tensor(edges, dtype=torch.long).permute(1, 0))
    return torch.cat([src_ids, tgt_ids], dim=1).permute(1, 0), data


def load_scene_info(scene, num_frames, size, config):
    """Same signature as cityscapes.py:79: `scene` is the path prefix of the per-instance track files (scene + "*.txt",
    glob order = node order, as in the reference)."""
    tracks = []
    for path in glob.glob(scene + "*.txt"):
        with open(path, "r") as f:
            tracks.append(f.read().splitlines())
    return scene_graph(tracks, size, config["train_params"]["num_input_frames"], num_frames,
                       config["test_params"]["lambda_traj"])


def tracking_mask(instance, tracking_ids):
    """cityscapes.py:43-50,196-199: 1 where the frame's instance id is one of the tracked ids of that frame.
    instance [T,H,W] (or [1,T,H,W]) integer ids, tracking_ids [T,N] -> [1,T,H,W] float32 (any device)."""
    inst = instance.reshape(instance.shape[-3:])
    ids = tracking_ids.to(inst.device)
    hit = (inst.unsqueeze(1).to(torch.int64) == ids[:, :, None, None]).any(1)
    return hit.to(torch.float32).unsqueeze(0)


def collate_graphs(graphs):
    """`Batch.from_data_list` for the fields above (train.py:31-32): node tensors concatenated along dim 0, edge_index
    along dim 1 with each scene's indices shifted by the nodes before it, `batch` = scene index of every node."""
    if not graphs:
        raise ValueError("empty batch")
    out, offset, edge, bvec = {}, 0, [], []
    for k in _NODE_FIELDS:
        if all(hasattr(g, k) for g in graphs):
            out[k] = torch.cat([getattr(g, k) for g in graphs], 0)
    for b, g in enumerate(graphs):
        edge.append(g.edge_index + offset)
        bvec.append(torch.full((g.num_nodes,), b, dtype=torch.long))
        offset += g.num_nodes
    ptr = torch.tensor([0] + [g.num_nodes for g in graphs]).cumsum(0)
    return GraphBatch(edge_index=torch.cat(edge, 1), batch=torch.cat(bvec), ptr=ptr, num_nodes=offset,
                      num_real_nodes=torch.cat([g.num_real_nodes for g in graphs]), **out)


def collate(samples):
    """train.py:23-38: tensors are stacked along a new batch axis, graphs are batched, anything else stays a list."""
    keys = sorted(set().union(*samples))
    out = {}
    for k in keys:
        v = [s.get(k) for s in samples]
        if torch.is_tensor(v[0]):
            out[k] = torch.stack(v, dim=0)
        elif isinstance(v[0], GraphData):
            out[k] = collate_graphs(v)
        else:
            out[k] = v
    return out


def read_flo(fn):
    """Middlebury .flo (utils.py:324-343): float32 magic 202021.25, int32 w, int32 h, then h*w*2 float32 (u, v
    interleaved), little-endian.  Returns [h,w,2] float32; a wrong magic prints a message and returns None, like the
    reference."""
    with open(fn, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or np.frombuffer(head, "<f4", 1)[0] != np.float32(202021.25):
            print("Magic number incorrect. Invalid .flo file")
            return None
        w, h = (int(v) for v in np.frombuffer(head, "<i4", 2, 4))
        data = np.frombuffer(f.read(8 * w * h), "<f4")
    return np.resize(data, (h, w, 2))
