"""c2m_amd: MI355X-native (gfx950) implementation of the C2M generator train-step hot path.

Drop-in surface: c2m_amd.modules.model.GeneratorFullModel and the block/loss classes under the reference's own
module paths.  `install_as_reference_layout()` registers them as top-level `modules` / `losses` and PATCHES the hot-path
functions (`resample`, `get_occlusion_map`, `resize_flow`, ...) into the reference's own `utils` package, which stays in
place with everything else the reference's src/train.py and Trainer import from it (Visualizer, save_parameters,
dist_all_gather_tensor, init_cudnn, ...).  Kernels: c2m_amd/csrc (HIP) behind include/c2m_hip.h.
"""
import importlib
import importlib.util
import sys
import types

__version__ = "0.2.0"

_MODULE_ALIASES = [
    "modules", "modules.model", "modules.layers", "modules.layers.down_block", "modules.layers.same_block",
    "modules.layers.up_block", "modules.layers.residual_block", "modules.layers.spade_block",
    "modules.layers.vgg", "modules.layers.utils", "modules.generator", "modules.generator.generator",
    "modules.generator.flowembedder", "modules.motion_estimator", "modules.motion_estimator.dense_motion",
    "modules.motion_estimator.motion_autoencoder", "modules.motion_estimator.sparse_encoder",
    "modules.motion_estimator.sparse_motion_estimator", "modules.appearance_encoder",
    "modules.appearance_encoder.appearance_encoder", "modules.discriminator",
    "modules.discriminator.discriminator", "modules.third_party", "modules.third_party.flow_net",
    "modules.third_party.flow_net.flow_net", "modules.third_party.flow_net.flownet2", "losses", "losses.losses"]

# hot-path functions of the reference's utils package (src/utils/ops.py:187-202,263-275, src/utils/utils.py:346-379)
# that are replaced by the HIP-backed ones; every other member of the reference's `utils` is left untouched
_UTILS_PATCHES = {"utils.ops": ("resample", "get_grid", "get_occlusion_map"),
                  "utils.utils": ("resize_flow", "resize_video", "isnan")}


def _reference_utils():
    """The reference's own `utils` package if it is importable (its src/ directory is on sys.path), else None."""
    mod = sys.modules.get("utils")
    if mod is not None:
        return None if getattr(mod, "__name__", "").startswith("c2m_amd") else mod
    try:
        spec = importlib.util.find_spec("utils")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    return importlib.import_module("utils")


def install_as_reference_layout():
    """Make the reference's `train.py` / `Trainer` run on this package without editing them.

    * `modules.*` and `losses.*` resolve to c2m_amd (same class names, constructor signatures, state_dict surface).
    * `utils`: when the reference's package is importable it is KEPT and only the hot-path functions listed in
      `_UTILS_PATCHES` are replaced inside `utils`, `utils.ops` and `utils.utils`; when it is not on sys.path (stand-alone
      use of the model code) `utils`, `utils.ops`, `utils.utils` alias c2m_amd.utils, which carries the helpers the
      trainer reads (`dist_all_reduce_tensor`, `dist_all_gather_tensor`, `init_cudnn`, `set_random_seed`, ...).
    * `torch_geometric.data.{Data,Batch}` fall back to the PyG-free attribute bags (`Batch.from_data_list`,
      `.pin_memory()`, `.to()` as train.py:23-38 and trainer.py:105-113 use them) if torch_geometric is not installed.
    Returns "patched" or "aliased" (what happened to `utils`)."""
    for n in _MODULE_ALIASES:
        sys.modules[n] = importlib.import_module("c2m_amd." + n)
    ours = importlib.import_module("c2m_amd.utils")
    ref = _reference_utils()
    if ref is not None:
        for sub, names in _UTILS_PATCHES.items():
            m = importlib.import_module(sub)
            for name in names:
                fn = getattr(ours, name)
                setattr(m, name, fn)
                setattr(ref, name, fn)           # `from .utils import *` / `from .ops import *` copies in utils/__init__
        mode = "patched"
    else:
        sys.modules["utils"] = ours
        sys.modules["utils.ops"] = importlib.import_module("c2m_amd.utils.ops")
        sys.modules["utils.utils"] = importlib.import_module("c2m_amd.utils.utils")
        mode = "aliased"
    if "torch_geometric" not in sys.modules and importlib.util.find_spec("torch_geometric") is None:
        from . import graph, synthetic
        tg = types.ModuleType("torch_geometric")
        tg.data = types.ModuleType("torch_geometric.data")
        tg.data.Batch, tg.data.Data = synthetic.GraphBatch, graph.GraphData
        sys.modules["torch_geometric"] = tg
        sys.modules["torch_geometric.data"] = tg.data
    return mode
