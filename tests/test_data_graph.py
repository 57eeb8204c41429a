"""Host side of the input pipeline: scene graphs from tracking files, PyG-free collation, .flo parsing.
Golden = outputs of the live reference (datasets/cityscapes.py::load_scene_info, utils.read_flow) on the committed
track files (oracle/capture_golden.py::capture_data); integer and float32 results must be bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

from c2m_amd import graph as G
from oracle import data_prep as D
from golden_io import Case, GOLDEN

TRACKS = os.path.join(GOLDEN, "scene_tracks")
FIELDS = {"x": "x", "source_frames_nodes_roi": "roi", "source_frames_nodes_roi_padded": "roi_pad",
          "target_frames_nodes_roi": "tgt_roi", "source_frames_nodes_instance_ids": "src_ids",
          "target_frames_nodes_instance_ids": "tgt_ids", "targets_barycenter": "bary", "y": "bary",
          "targets_displacement": "disp", "targets_theta": "theta"}


@pytest.fixture(scope="module")
def case():
    return Case("data_scene_graph")


def _tracks(prefix):
    return [open(p).read().splitlines() for p in sorted(glob.glob(os.path.join(TRACKS, prefix) + "*.txt"))]


def test_oracle_scene_info_matches_the_reference(case):
    for c in case.meta["cases"]:
        o = D.scene_info(_tracks(c["prefix"]), (128, 256), c["t_in"], 7, c["lambda_traj"])
        order = np.argsort(o["src_ids"][:, 0])
        for ref_name, ours in FIELDS.items():
            want = case.arr[f"{c['tag']}.{ref_name}"]
            assert o[ours].dtype == want.dtype and np.array_equal(o[ours][order], want), (c["tag"], ref_name)
        assert np.array_equal(o["edge_index"], case.arr[c["tag"] + ".edge_index"])
        assert np.array_equal(o["tracking_ids"][:, order], case.arr[c["tag"] + ".tracking_ids"])


def test_load_scene_info_matches_the_reference(case):
    for c in case.meta["cases"]:
        cfg = {"train_params": {"num_input_frames": c["t_in"]}, "test_params": {"lambda_traj": c["lambda_traj"]}}
        ids, d = G.load_scene_info(os.path.join(TRACKS, c["prefix"]), 7, [128, 256], cfg)
        order = torch.argsort(d.source_frames_nodes_instance_ids[:, 0])
        for ref_name in FIELDS:
            want = torch.from_numpy(case.arr[f"{c['tag']}.{ref_name}"])
            got = getattr(d, ref_name)[order]
            assert got.dtype == want.dtype and torch.equal(got, want), (c["tag"], ref_name)
        assert torch.equal(d.edge_index, torch.from_numpy(case.arr[c["tag"] + ".edge_index"]))
        assert torch.equal(d.num_real_nodes, torch.from_numpy(case.arr[c["tag"] + ".num_real_nodes"]))
        assert d.num_real_nodes.dtype == torch.int32
        assert torch.equal(ids[:, order], torch.from_numpy(case.arr[c["tag"] + ".tracking_ids"]))


def test_scene_graph_error_behaviour():
    line = "100,100,50,40,0.9,%d"
    with pytest.raises(IndexError):                       # np.eye(19)[26] in the reference
        G.scene_graph([[line % 26001] * 7], (128, 256), 2, 7)
    with pytest.raises(ValueError):
        G.scene_graph([], (128, 256), 2, 7)
    with pytest.raises(ValueError):
        G.scene_graph([[line % 11000] * 3], (128, 256), 2, 7)


def test_read_flo(case, capsys):
    for impl in (G.read_flo, D.read_flo):
        flo = impl(os.path.join(TRACKS, "tiny.flo"))
        assert flo.dtype == np.float32 and np.array_equal(flo, case.arr["flo.tiny"])
    assert G.read_flo(os.path.join(TRACKS, "bad_magic.flo")) is None
    assert "Magic number incorrect" in capsys.readouterr().out


def test_collate_graphs_is_batch_from_data_list():
    cfg = {"train_params": {"num_input_frames": 2}, "test_params": {"lambda_traj": 1}}
    _, a = G.load_scene_info(os.path.join(TRACKS, "aachen_000000_000019_"), 7, [128, 256], cfg)
    _, b = G.load_scene_info(os.path.join(TRACKS, "bonn_000001_000004_"), 7, [128, 256], cfg)
    batch = G.collate_graphs([a, b, a])
    assert batch.num_nodes == 7 and batch.batch.tolist() == [0, 0, 0, 1, 2, 2, 2]
    assert batch.num_real_nodes.tolist() == [3, 1, 3] and batch.ptr.tolist() == [0, 3, 4, 7]
    assert torch.equal(batch.x, torch.cat([a.x, b.x, a.x])) and batch.targets_theta.shape == (7, 5, 6)
    e = batch.edge_index
    assert e.shape == (2, 6 + 1 + 6) and torch.equal(e[:, :6], a.edge_index)
    assert e[:, 6].tolist() == [3, 3]                     # the single-node scene keeps its [0, 0] self edge, shifted
    assert torch.equal(e[:, 7:], a.edge_index + 4)
    moved = batch.to("cpu")                               # the attribute bag the model consumes
    assert moved.source_frames_nodes_roi_padded.shape == (7, 2, 4)


def test_collate_samples_and_tracking_mask():
    cfg = {"train_params": {"num_input_frames": 2}, "test_params": {"lambda_traj": 1}}
    ids, g = G.load_scene_info(os.path.join(TRACKS, "aachen_000000_000019_"), 7, [128, 256], cfg)
    gen = torch.Generator().manual_seed(0)
    pool = torch.tensor([0, 11003, 13001, 18002, 24000], dtype=torch.int32)
    inst = pool[torch.randint(0, 5, (1, 7, 16, 32), generator=gen)]
    mask = G.tracking_mask(inst, ids)
    want = torch.zeros(1, 7, 16, 32)
    for t in range(7):                                    # cityscapes.py:43-50
        for i in ids[t].tolist():
            want[0, t] = torch.where(inst[0, t] == i, torch.ones(16, 32), want[0, t])
    assert torch.equal(mask, want) and 0 < mask.mean() < 1
    sample = {"video": torch.zeros(3, 7, 16, 32), "tracking_gnn": g, "tracking_mask": mask, "complete_list": ["a", "b"]}
    out = G.collate([sample, sample])
    assert out["video"].shape == (2, 3, 7, 16, 32) and out["tracking_gnn"].num_nodes == 6
    assert out["complete_list"] == [["a", "b"], ["a", "b"]]


# ------------------------------------------------------------------------ image / mask half of the dataset code (round 3)
def test_oracle_dataset_prep_matches_the_reference():
    """oracle/data_prep.py vs the tensors the LIVE reference built (datasets/cityscapes.py:20-70,195-265, run by
    oracle/capture_golden.py::capture_dataset on PNG / .flo files written from the stored arrays): bit-exact.  This pins the
    oracle the GPU kernels of data_prep.hip are held to."""
    c = Case("data_dataset_prep")
    i, o = c.group("in"), c.group("out")
    assert torch.equal(D.read_video(i["frames"].numpy()), o["video"])
    bg, fg = D.read_seg_masks(i["labels"].numpy())
    assert torch.equal(bg, o["bg_mask"]) and torch.equal(fg, o["fg_mask"])
    assert float(bg.sum() + fg.sum()) == float((i["labels"] < 20).sum())
    assert torch.equal(i["inst"].unsqueeze(0), o["instance_mask"])                       # ToTensor of an integer array: ids as they are
    occ, flow = D.load_flow_occ(i["occ"][1:].numpy(), i["flow"][1:].numpy())             # frames 1..T-1 (cityscapes.py:239)
    assert torch.equal(occ, o["target_bw_occ"]) and torch.equal(flow, o["target_bw_of"])
    assert set(o["target_bw_occ"].unique().tolist()) <= {0.0, 1.0}
    # tracking mask: the ids of the committed track files, per frame
    tracks = _tracks(c.meta["track_prefix"])
    info = D.scene_info(tracks, (c.meta["H"], c.meta["W"]), 2, c.meta["T"])
    tm = G.tracking_mask(i["inst"], torch.from_numpy(info["tracking_ids"]))
    assert torch.equal(tm, o["tracking_mask"].unsqueeze(0) if o["tracking_mask"].dim() == 3 else o["tracking_mask"])
    assert 0 < float(tm.sum()) < float((i["inst"] > 0).sum())                            # the untracked instance stays 0
