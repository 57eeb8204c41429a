"""c2m_amd: MI355X-native (gfx950) implementation of the C2M generator train-step hot path.

Drop-in surface: c2m_amd.modules.model.GeneratorFullModel and the block/loss classes under the reference's own
module paths (`install_as_reference_layout()` registers them as top-level `modules`, `losses`, `utils` so that the
reference's src/train.py and Trainer import them unchanged).  Kernels: c2m_amd/csrc (HIP) behind include/c2m_hip.h.
"""
import importlib
import sys

__version__ = "0.1.0"


def install_as_reference_layout():
    """Alias c2m_amd.{modules,losses,utils} (and their submodules) under the reference's import names."""
    names = ["modules", "modules.model", "modules.layers", "modules.layers.down_block", "modules.layers.same_block",
             "modules.layers.up_block", "modules.layers.residual_block", "modules.layers.spade_block",
             "modules.layers.vgg", "modules.layers.utils", "modules.generator", "modules.generator.generator",
             "modules.generator.flowembedder", "modules.motion_estimator", "modules.motion_estimator.dense_motion",
             "modules.motion_estimator.motion_autoencoder", "modules.motion_estimator.sparse_encoder",
             "modules.motion_estimator.sparse_motion_estimator", "modules.appearance_encoder",
             "modules.appearance_encoder.appearance_encoder", "modules.discriminator",
             "modules.discriminator.discriminator", "modules.third_party", "modules.third_party.flow_net",
             "modules.third_party.flow_net.flow_net", "losses", "losses.losses", "utils", "utils.ops", "utils.utils"]
    for n in names:
        sys.modules[n] = importlib.import_module("c2m_amd." + n)
    from . import synthetic
    import types
    tg = types.ModuleType("torch_geometric")
    tg.data = types.ModuleType("torch_geometric.data")
    tg.data.Batch = tg.data.Data = synthetic.GraphBatch
    sys.modules.setdefault("torch_geometric", tg)
    sys.modules.setdefault("torch_geometric.data", tg.data)
