"""Drop-in boundary: the reference's OWN src/train.py and trainer.trainer.Trainer, imported unchanged from
/root/reference/src, drive c2m_amd's GeneratorFullModel after `c2m_amd.install_as_reference_layout()`.

Build-container only (the reference does not travel to the GPU box): skipped when /root/reference is absent.  Runs in a
child process because it re-binds the top-level names `modules`, `losses`, `utils`.  Third-party packages that the
reference imports but this image lacks (imageio, cv2, torchvision, dominate, tensorboard ...) are replaced by EMPTY
stand-ins for the import only; nothing of theirs is on the path under test.
"""
import os
import subprocess
import sys
import textwrap

import pytest

REF_SRC = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference tree not present (GPU box)")

_CHILD = r'''
import copy, importlib, importlib.machinery, sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


class _Anything(types.ModuleType):
    """import-only stand-in: any attribute is another stand-in / a no-op callable"""
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []
        self.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Anything(self.__name__ + "." + k)
        sys.modules[m.__name__] = m
        setattr(self, k, m)
        return m
    def __call__(self, *a, **k):
        return None


class _Finder:
    MISSING = ("imageio", "cv2", "torchvision", "dominate", "tensorboardX", "tensorboard", "skimage", "lpips",
               "tensorflow", "tensorflow_hub", "tensorflow_gan", "pycocotools")
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
    def create_module(self, spec):
        return _Anything(spec.name)
    def exec_module(self, module):
        pass

sys.meta_path.append(_Finder())
import scipy
if not hasattr(scipy, "misc"):
    sys.modules["scipy.misc"] = scipy.misc = _Anything("scipy.misc")

import torch
sys.path.insert(0, REF_SRC)
import c2m_amd
mode = c2m_amd.install_as_reference_layout()
assert mode == "patched", mode

# --- the reference's own utils package is still the reference's, with only the hot-path functions replaced
import utils, utils.ops, utils.utils
assert utils.__file__.startswith(REF_SRC), utils.__file__
from utils.visualizer import Visualizer                      # trainer/base.py:2
from utils.utils import set_random_seed, init_cudnn          # train.py:11
import c2m_amd.utils as ours
for name in ("resample", "get_grid", "get_occlusion_map", "resize_flow", "resize_video", "isnan"):
    assert getattr(utils, name) is getattr(ours, name), name
assert utils.ops.resample is ours.resample and utils.utils.resize_flow is ours.resize_flow
for name in ("dist_all_reduce_tensor", "dist_all_gather_tensor", "save_parameters", "tensor2im", "grid_sample",
             "get_corresponding_map"):
    assert getattr(utils, name).__module__.startswith("utils."), name     # untouched reference code

# --- train.py imports unchanged (train.py:1-19) and finds our classes behind the reference's names
import train
from trainer.trainer import Trainer
import c2m_amd.modules.model as ours_model
assert train.GeneratorFullModel is ours_model.GeneratorFullModel
from c2m_amd.graph import GraphData
from c2m_amd.synthetic import GraphBatch, make_batch
assert train.Data is GraphData and train.Batch is GraphBatch

# --- construct exactly as train.py:71-98 does
from c2m_amd.config import default_config, normalize_config
cfg = normalize_config(default_config(num_input_frames=1, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16,
                                      out_channel=16, ndf=4, height=32, width=64))
cfg["train_params"]["use_pre_processed_of"] = True
cfg["train_params"]["continue_train"] = False
cfg["train_params"].setdefault("batch_size", 2)
cfg.setdefault("visualizer_params", {"display_freq": 100, "print_freq": 100})
cfg.setdefault("checkpoint_params", {})
cfg.setdefault("name", "dropin_test")
cfg.setdefault("dataset_params", {})["dataset"] = "cityscapes"
cfg["train_params"].setdefault("eval_freq", 100)
set_random_seed(0, by_rank=False)
c2m = train.GeneratorFullModel(train_params=cfg["train_params"], model_params=cfg["model_params"],
                               dataset=cfg["dataset_params"]["dataset"])
dev = torch.device("cpu")            # no GPU in the build container; `local_rank == 0` (Visualizer, disk writes) is False
c2m.to(dev)
tp = cfg["train_params"]
optimizer_vae, optimizer_gnn = c2m.optimizer, c2m.optimizer_gnn                      # train.py:88-95
scheduler_vae, scheduler_gnn = c2m.scheduler_g, c2m.scheduler_gnn
scheduler_d_image = c2m.scheduler_d_image if tp["use_image_discriminator"] else None
optimizer_d_image = c2m.d_optimizer_image if tp["use_image_discriminator"] else None
scheduler_d_video = c2m.scheduler_d_video if tp["use_video_discriminator"] else None
optimizer_d_video = c2m.d_optimizer_video if tp["use_video_discriminator"] else None
for o in (optimizer_vae, optimizer_gnn, optimizer_d_image, optimizer_d_video):
    assert isinstance(o, torch.optim.Adam)
opt = types.SimpleNamespace(device_ids=[0], seed=0, profile=False, config="x")
loader = [None] * 4
trainer = Trainer(cfg, opt, c2m, None, optimizer_vae, optimizer_gnn, optimizer_d_image, optimizer_d_video,
                  scheduler_vae, scheduler_gnn, scheduler_d_image, scheduler_d_video, loader, loader, dev)
start_epoch, epoch_iter = trainer.load_checkpoint()
trainer.initialize_deltas(start_epoch, epoch_iter)
trainer.start_of_epoch(start_epoch)

# --- one batch through train.py's own BatchCollate (Batch.from_data_list on GraphData) and pin_memory surface
full = make_batch(2, 32, 64, 1, seed=0)
g = full["tracking_gnn"]
samples = []
for b in range(2):
    sel = (g.batch == b).nonzero().flatten()
    n0, n = int(sel[0]), int(sel.numel())
    emask = (g.edge_index[0] >= n0) & (g.edge_index[0] < n0 + n)
    gd = GraphData(x=g.x[sel], targets_theta=g.targets_theta[sel], edge_index=g.edge_index[:, emask] - n0,
                   num_real_nodes=g.num_real_nodes[b:b + 1],
                   source_frames_nodes_roi_padded=g.source_frames_nodes_roi_padded[sel],
                   source_frames_nodes_instance_ids=g.source_frames_nodes_instance_ids[sel])
    s = {k: v[b] for k, v in full.items() if torch.is_tensor(v)}
    s["tracking_gnn"] = gd
    s["complete_list"] = "clip%d" % b
    samples.append(s)
batch = train.collate_wrapper(samples)
assert isinstance(batch.data["tracking_gnn"], GraphBatch)
assert torch.equal(batch.data["tracking_gnn"].edge_index, g.edge_index)
assert torch.equal(batch.data["tracking_gnn"].batch, g.batch)
assert hasattr(batch.data["tracking_gnn"], "pin_memory")
data = trainer.start_of_iteration(batch)                     # trainer.py:100-115: .to(local_rank) on every value
assert data["input_of"] is None and data["video"].shape == (2, 3, 6, 32, 64)

# --- update_model runs the reference's code up to the first HIP op, which refuses CPU tensors loudly (no fallback)
try:
    trainer.update_model(data)
except RuntimeError as e:
    assert "HIP device" in str(e), e
    import traceback
    tb = traceback.extract_tb(e.__traceback__)
    files = [f.filename for f in tb]
    assert any(f.startswith(REF_SRC + "/trainer/trainer.py") for f in files), files
    assert any("/c2m_amd/ops.py" in f for f in files), files
else:
    raise AssertionError("update_model ran without a GPU: a CPU fallback exists on the product path")
print("DROPIN-OK")
'''


def test_reference_train_py_and_trainer_drive_c2m_amd_unchanged():
    code = f"ROOT = {ROOT!r}\nREF_SRC = {REF_SRC!r}\n" + textwrap.dedent(_CHILD)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]


def test_layout_without_reference_aliases_utils():
    """Stand-alone use (no reference on sys.path): `utils` resolves to c2m_amd.utils with the trainer-facing helpers."""
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import c2m_amd
        assert c2m_amd.install_as_reference_layout() == "aliased"
        import utils
        from utils.utils import set_random_seed, init_cudnn
        for n in ("resample", "get_occlusion_map", "resize_flow", "resize_video", "dist_all_reduce_tensor",
                  "dist_all_gather_tensor", "get_world_size", "is_master"):
            assert hasattr(utils, n), n
        from torch_geometric.data import Batch, Data
        assert hasattr(Batch, "from_data_list") and hasattr(Batch, "pin_memory")
        import torch
        t = torch.ones(3)
        assert utils.dist_all_gather_tensor(t) is t and utils.dist_all_reduce_tensor(t) is t
        print("ALIAS-OK")
    """)
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALIAS-OK" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
