"""Synthetic Cityscapes-shaped batches (SURVEY.md §8d "Synthetic inputs").

Produces exactly the batch-dict contract the reference model consumes
(src/datasets/cityscapes.py:301-326, collated by src/train.py:23-38):  video, bg_mask, fg_mask,
instance_mask, tracking_gnn, target_bw_of, target_bw_occ, input_of, input_occ.  `tracking_gnn`
is a plain attribute bag (the model only does attribute access, so no torch_geometric is needed).

All draws use a CPU torch.Generator so the same seed gives the same batch on any host/device.
"""
import torch


class GraphBatch:
    """Attribute bag standing in for torch_geometric.data.Batch (see module docstring)."""

    _tensor_fields = ("x", "targets_theta", "edge_index", "batch", "num_real_nodes",
                      "source_frames_nodes_roi_padded", "source_frames_nodes_instance_ids")

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device, non_blocking=False):
        out = GraphBatch(**self.__dict__)
        for k in self._tensor_fields:
            setattr(out, k, getattr(self, k).to(device, non_blocking=non_blocking))
        return out

    def clone(self):
        out = GraphBatch(**self.__dict__)
        for k in self._tensor_fields:
            setattr(out, k, getattr(self, k).clone())
        return out

    def pin_memory(self):
        """train.py:34-37 pins every collated value; tensors of any field are pinned, the rest is kept."""
        import torch
        out = GraphBatch(**self.__dict__)
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                setattr(out, k, v.pin_memory())
        return out

    @classmethod
    def from_data_list(cls, data_list):
        """`torch_geometric.data.Batch.from_data_list` as train.py:31-32 calls it (c2m_amd.graph.collate_graphs)."""
        from .graph import collate_graphs
        return collate_graphs(list(data_list))


def make_batch(batch_size=1, height=128, width=256, num_input_frames=2, num_predicted_frames=5,
               num_objects=3, seed=0, device="cpu", use_fw_of=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    B, H, W = batch_size, height, width
    T = num_input_frames + num_predicted_frames
    sy, sx = H / 128.0, W / 256.0

    video = torch.rand(B, 3, T, H, W, generator=g)
    # 8x8-blocky semantic ids in [0,20) -> one-hot, channels 0..10 = bg, 11..19 = fg
    bh, bw = max(H // 8, 1), max(W // 8, 1)
    sem = torch.randint(0, 20, (B, 1, T, bh, bw), generator=g)
    sem = sem.repeat_interleave(H // bh, dim=3).repeat_interleave(W // bw, dim=4)
    onehot = torch.zeros(B, 20, T, H, W).scatter_(1, sem, 1.0)
    bg_mask, fg_mask = onehot[:, :11].contiguous(), onehot[:, 11:].contiguous()

    instance = torch.zeros(B, 1, T, H, W, dtype=torch.int32)
    xs, thetas, rois, ids, batch_vec, edges = [], [], [], [], [], []
    node0 = 0
    per_sample = [int(num_objects)] * B if isinstance(num_objects, int) else [int(n) for n in num_objects]
    if len(per_sample) != B or min(per_sample) < 1:
        raise ValueError("num_objects: an int or one count >= 1 per sample (ragged graphs are allowed)")
    for b in range(B):
        num_objects = per_sample[b]
        for n in range(num_objects):
            x0 = int(round((20 + 60 * n) * sx))
            y0 = int(round((40 + 10 * n) * sy))
            bw_, bh_ = max(int(round(40 * sx)), 2), max(int(round(30 * sy)), 2)
            x1, y1 = min(x0 + bw_, W), min(y0 + bh_, H)
            inst_id = 26001 + n
            instance[b, 0, :, y0:y1, x0:x1] = inst_id
            cy = ((y0 + y1) / 2.0) / H * 2 - 1
            cx = ((x0 + x1) / 2.0) / W * 2 - 1
            cls = torch.zeros(19)
            cls[inst_id // 1000 % 19] = 1.0
            feat = torch.cat([torch.tensor([cy, cx, (y1 - y0) / H, (x1 - x0) / W]), cls])
            xs.append(feat.unsqueeze(0).repeat(num_input_frames, 1))
            thetas.append(torch.tensor([[1.0, 0.0, 0.02 * (t + 1), 0.0, 1.0, 0.01 * (t + 1)]
                                        for t in range(num_predicted_frames)]))
            roi = torch.tensor([max(x0 - 15 * sx, 0.0), min(x1 + 15 * sx, float(W)),
                                max(y0 - 10 * sy, 0.0), min(y1 + 10 * sy, float(H))])
            rois.append(roi.unsqueeze(0).repeat(num_input_frames, 1))
            ids.append(torch.full((num_input_frames,), inst_id, dtype=torch.long))
            batch_vec.append(b)
        for i in range(num_objects):
            for j in range(num_objects):
                if i != j:
                    edges.append([node0 + i, node0 + j])
        node0 += num_objects
    if not edges:
        edges = [[0, 0]]
    tracking_gnn = GraphBatch(
        x=torch.stack(xs, 0), targets_theta=torch.stack(thetas, 0),
        edge_index=torch.tensor(edges, dtype=torch.long).t().contiguous(),
        batch=torch.tensor(batch_vec, dtype=torch.long),
        num_real_nodes=torch.tensor(per_sample, dtype=torch.int32),
        num_nodes=sum(per_sample),
        source_frames_nodes_roi_padded=torch.stack(rois, 0),
        source_frames_nodes_instance_ids=torch.stack(ids, 0))

    target_bw_of = 2.0 * torch.randn(B, 2, num_predicted_frames, H, W, generator=g)
    target_bw_occ = (torch.rand(B, 1, num_predicted_frames, H, W, generator=g) > 0.2).float()
    batch = dict(video=video, bg_mask=bg_mask, fg_mask=fg_mask, instance_mask=instance,
                 tracking_gnn=tracking_gnn, target_bw_of=target_bw_of, target_bw_occ=target_bw_occ,
                 input_of=None, input_occ=None)
    if num_input_frames > 1:
        batch["input_of"] = 2.0 * torch.randn(B, 2, num_input_frames - 1, H, W, generator=g)
        batch["input_occ"] = (torch.rand(B, 1, num_input_frames - 1, H, W, generator=g) > 0.2).float()
    if use_fw_of:
        # forward flow / occlusion targets of the `use_fw_of: True` branch (dense_motion.py:71-87, losses.py:211-216): drawn from
        # their OWN generator so that the tensors above keep the values every older fixture was captured with
        g2 = torch.Generator(device="cpu").manual_seed(seed + 7919)
        batch["target_fw_of"] = 2.0 * torch.randn(B, 2, num_predicted_frames, H, W, generator=g2)
        batch["target_fw_occ"] = (torch.rand(B, 1, num_predicted_frames, H, W, generator=g2) > 0.2).float()
    return batch_to(batch, device)


def make_stream_batch(streams=1, windows=2, height=128, width=256, num_input_frames=2, num_predicted_frames=5,
                      num_objects=3, seed=0, device="cpu", use_fw_of=False):
    """BASELINE configs[4] ("14-frame clips"): t_out is fixed at 5 by the model (UpBlock2d chunk(5, 0)),
    so a 14-frame stream sample is cut into `windows` consecutive 7-frame windows that are stacked along the batch axis
     -- window w of stream s is clip s * windows + w.  Every synthetic field is drawn i.i.d. per frame (and
    the instance rectangles are static), so cutting one long draw into windows and drawing the windows directly are the
    same distribution; the windows are drawn directly."""
    if windows < 1 or streams < 1:
        raise ValueError("streams and windows must be >= 1")
    return make_batch(streams * windows, height, width, num_input_frames, num_predicted_frames, num_objects, seed, device, use_fw_of)


def batch_to(batch, device):
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device) if v is not None else None
    return out


def make_step_rng(batch, z_dim, latent_dim=1024, num_predicted_frames=5, seed=0):
    """The three random draws of one training forward, made explicit so that parity runs can inject them.

    Reference: latent_traj ~ N(0,1) on CPU (model.py:157-160), eps = randn_like(std) (dense_motion.py:90),
    one np.random click index per sample (sparse_motion_estimator.py:46-51).
    """
    g = torch.Generator(device="cpu").manual_seed(10_000 + seed)
    gnn = batch["tracking_gnn"]
    N = gnn.x.shape[0]
    B = batch["video"].shape[0]
    latent_traj = torch.randn(N, num_predicted_frames, z_dim, generator=g)
    eps = torch.randn(B, latent_dim, generator=g)
    clicks, total = [], 0
    for n in gnn.num_real_nodes.tolist():
        clicks.append(int(torch.randint(0, int(n), (1,), generator=g)) + total)
        total += int(n)
    return dict(latent_traj=latent_traj, eps=eps, click_index=torch.tensor(clicks, dtype=torch.long))
