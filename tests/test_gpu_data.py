"""Device-side batch assembly vs the CPU restatement of the reference's dataset arithmetic: bit-exact."""
import numpy as np
import pytest
import torch

from c2m_amd import data
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch
from oracle import data_prep as D

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _arrays(B, T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 34, (B, T, H, W), generator=g, dtype=torch.uint8)       # ids >= 20 belong to no channel
    n = min(20, W)
    labels[0, 0, 0, :n] = torch.arange(n, dtype=torch.uint8)
    occ = (torch.randint(0, 2, (B, T, H, W), generator=g) * 255).to(torch.uint8)
    occ[0, 0, 0, :4] = torch.tensor([127, 128, 0, 255], dtype=torch.uint8)            # around the 0.5 threshold (W >= 4)
    flow = torch.randn(B, T, H, W, 2, generator=g) * 3.0
    return frames, labels, occ, flow


@pytest.mark.parametrize("shape", [(2, 7, 16, 32), (1, 1, 5, 7), (3, 5, 33, 17)])
def test_prep_kernels_bitexact_vs_oracle(shape):
    B, T, H, W = shape
    frames, labels, occ, flow = _arrays(B, T, H, W, 5)
    video = data.prep_video(frames.to(DEV)).cpu()
    bg, fg = (t.cpu() for t in data.prep_seg_onehot(labels.to(DEV)))
    o, f = (t.cpu() for t in data.prep_flow_occ(occ.to(DEV), flow.to(DEV)))
    for b in range(B):
        assert torch.equal(video[b], D.read_video(frames[b].numpy()))
        rbg, rfg = D.read_seg_masks(labels[b].numpy())
        assert torch.equal(bg[b], rbg) and torch.equal(fg[b], rfg)
        ro, rf = D.load_flow_occ(occ[b].numpy(), flow[b].numpy())
        assert torch.equal(o[b], ro) and torch.equal(f[b], rf)
    assert float(bg.sum() + fg.sum()) == float((labels < 20).sum()), "exactly one channel per pixel with an id < 20"


def test_prep_rejects_wrong_inputs_and_empty_batch():
    with pytest.raises(TypeError):
        data.prep_video(torch.zeros(1, 1, 4, 4, 3, device=DEV))
    with pytest.raises(RuntimeError):
        data.prep_video(torch.zeros(1, 1, 4, 4, 3, dtype=torch.uint8))          # not on the device: no CPU path
    assert data.prep_video(torch.zeros(0, 7, 4, 4, 3, dtype=torch.uint8, device=DEV)).shape == (0, 3, 7, 4, 4)


def test_assembled_batch_drives_the_model():
    """A batch assembled on the device from uint8 arrays feeds GeneratorFullModel.forward like the synthetic one."""
    B, T, H, W = 1, 7, 128, 256
    frames, labels, occ, flow = _arrays(B, T, H, W, 11)
    labels = labels % 20
    ref = make_batch(B, H, W, 2, seed=2)
    gnn = ref["tracking_gnn"]
    inst = ref["instance_mask"][:, 0]
    batch = data.assemble_batch(frames.to(DEV), labels.to(DEV), inst.to(DEV), occ[:, 2:].to(DEV), flow[:, 2:].to(DEV),
                                gnn.to(DEV) if hasattr(gnn, "to") else gnn, occ[:, 1:2].to(DEV), flow[:, 1:2].to(DEV))
    for k in ("video", "bg_mask", "fg_mask", "instance_mask", "target_bw_of", "target_bw_occ", "input_of", "input_occ"):
        assert tuple(batch[k].shape) == tuple(ref[k].shape) and batch[k].dtype == ref[k].dtype, k
    cfg = normalize_config(default_config(num_input_frames=2, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16,
                                          out_channel=16, ndf=4, use_image_discriminator=False,
                                          use_video_discriminator=False))
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=cfg["train_params"], model_params=cfg["model_params"], dataset="cityscapes")
    model.to(DEV).train()
    out, lg, _, _ = model(batch)
    assert all(np.isfinite(float(v.detach()) if torch.is_tensor(v) else float(v)) for v in lg.values())
    assert out["generated"].shape == (B, 3, 5, H, W)


def test_track_file_graphs_drive_the_model():
    """Scene graphs parsed from the committed tracking files, collated without torch_geometric (ragged: 3 + 1 objects),
    feed the training forward + backward; the tracked instances are painted from their ROIs."""
    import os
    from c2m_amd import graph as G
    from golden_io import GOLDEN
    B, T, H, W = 2, 7, 128, 256
    cfg_d = {"train_params": {"num_input_frames": 2}, "test_params": {"lambda_traj": 1}}
    graphs, ids = [], []
    for prefix in ("aachen_000000_000019_", "bonn_000001_000004_"):
        i, g = G.load_scene_info(os.path.join(GOLDEN, "scene_tracks", prefix), T, [H, W], cfg_d)
        graphs.append(g); ids.append(i)
    inst = torch.zeros(B, T, H, W, dtype=torch.int32)
    for b, g in enumerate(graphs):
        for n in range(g.num_nodes):
            x_l, x_r, y_t, y_b = (int(round(float(v))) for v in g.source_frames_nodes_roi[n, -1])
            inst[b, :, max(y_t, 0):min(y_b, H), max(x_l, 0):min(x_r, W)] = int(g.source_frames_nodes_instance_ids[n, -1])
    tmask = torch.stack([G.tracking_mask(inst[b].to(DEV), ids[b]) for b in range(B)])
    assert tmask.shape == (B, 1, T, H, W) and 0 < float(tmask.mean()) < 1
    frames, labels, occ, flow = _arrays(B, T, H, W, 13)
    batch = data.assemble_batch(frames.to(DEV), (labels % 20).to(DEV), inst.to(DEV), occ[:, 2:].to(DEV),
                                flow[:, 2:].to(DEV), G.collate_graphs(graphs).to(DEV), occ[:, 1:2].to(DEV),
                                flow[:, 1:2].to(DEV))
    cfg = normalize_config(default_config(num_input_frames=2, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16,
                                          out_channel=16, ndf=4, use_image_discriminator=False,
                                          use_video_discriminator=False))
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=cfg["train_params"], model_params=cfg["model_params"], dataset="cityscapes")
    model.to(DEV).train()
    out, lg, _, _ = model(batch)
    total = sum(v for v in lg.values() if torch.is_tensor(v))
    total.backward()
    assert np.isfinite(float(total)) and out["generated"].shape == (B, 3, 5, H, W)
    assert float(out["sparse_motion_bin"].sum()) > 0, "the rasteriser found the tracked instances"


def test_prep_kernels_bitexact_vs_reference_capture():
    """data_prep.hip vs the LIVE reference's dataset code (tests/golden/data_dataset_prep.npz, captured from
    datasets/cityscapes.py:20-70,195-265 on files written from the stored arrays): video, one-hot split, instance ids,
    occlusion clip and .flo layout bit-exact, and the tracking mask built on the device."""
    from golden_io import Case
    from c2m_amd import graph as G
    c = Case("data_dataset_prep")
    i, o = c.group("in"), c.group("out")
    dev = lambda t: t.unsqueeze(0).to(DEV)
    batch = data.assemble_batch(dev(i["frames"]), dev(i["labels"]), dev(i["inst"]), dev(i["occ"][1:]), dev(i["flow"][1:]), None)
    for k in ("video", "bg_mask", "fg_mask", "instance_mask", "target_bw_occ", "target_bw_of"):
        assert torch.equal(batch[k][0].cpu(), o[k]), k
    import glob, os
    from golden_io import GOLDEN
    tracks = [open(p).read().splitlines() for p in sorted(glob.glob(os.path.join(GOLDEN, "scene_tracks", c.meta["track_prefix"]) + "*.txt"))]
    ids, _ = G.scene_graph(tracks, (c.meta["H"], c.meta["W"]), 2, c.meta["T"])
    tm = G.tracking_mask(batch["instance_mask"][0], torch.as_tensor(ids))
    want = o["tracking_mask"]
    assert torch.equal(tm.cpu().reshape(want.shape), want)
