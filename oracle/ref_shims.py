"""TEST INFRASTRUCTURE -- runs ONLY in the build container, never on the GPU box.

Makes the read-only reference under /root/reference/src importable on CPU so that
oracle/capture_golden.py can generate golden vectors from the reference's own code.

What is injected (nothing from the reference is copied):
  * empty stand-ins for modules the reference imports but the hot path never calls
    (imageio, cv2), and thin module objects for torchvision / torch_geometric whose only
    used members are backed by OUR restatements in oracle/thirdparty.py
    (roi_align, GATv2Conv, vgg19 -- parity for those three is "unpinned", see thirdparty.py);
  * two CPU patches for hard-coded `.cuda()` calls: utils.ops.get_grid (ops.py:202) and
    GANLoss.get_target_tensor (discriminator.py:114,120).
"""
import math
import sys
import types

import torch
import torch.nn as nn

from . import thirdparty

REF_SRC = "/root/reference/src"


class _GATv2Conv(nn.Module):
    """Parameter container with PyG's GATv2Conv parameter names; forward = thirdparty.gatv2_conv."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2,
                 add_self_loops=True, **_):
        super().__init__()
        assert not concat and not add_self_loops
        self.heads, self.out_channels, self.negative_slope = heads, out_channels, negative_slope
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, heads * out_channels, bias=True)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for w in (self.lin_l.weight, self.lin_r.weight, self.att):  # glorot, as PyG
            a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
            w.data.uniform_(-a, a)
        self.lin_l.bias.data.zero_()
        self.lin_r.bias.data.zero_()

    def forward(self, x, edge_index):
        return thirdparty.gatv2_conv(x, edge_index, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight,
                                     self.lin_r.bias, self.att, self.bias, self.heads, self.negative_slope)


class _MessagePassing(nn.Module):
    def __init__(self, aggr="add", **_):
        super().__init__()


class _Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _vgg19(pretrained=False, **_):
    layers = []
    for _, kind, cin, cout in thirdparty.vgg19_feature_layout():
        if kind == "conv":
            conv = nn.Conv2d(cin, cout, 3, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            layers.append(conv)
        elif kind == "relu":
            layers.append(nn.ReLU(inplace=True))
        else:
            layers.append(nn.MaxPool2d(2, 2))
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


def _mod(name, **members):
    m = types.ModuleType(name)
    m.__dict__.update(members)
    sys.modules[name] = m
    return m


def install():
    """Inject stubs, put the reference on sys.path, import it and apply the CPU patches."""
    sys.dont_write_bytecode = True
    for empty in ("imageio", "cv2"):
        if empty not in sys.modules:
            _mod(empty)

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _ToTensor:
        """torchvision.transforms.ToTensor (absent third-party dependency; published behaviour restated, "unpinned"):
        PIL image -> ndarray (HWC), ndarray [H,W] -> [H,W,1]; CHW tensor; uint8 -> float32 / 255, other dtypes unchanged.
        Only capture_dataset() calls it -- what it pins is the reference's OWN code around it (cityscapes.py:20-70,195-265)."""

        def __call__(self, pic):
            import numpy as np
            arr = pic if isinstance(pic, np.ndarray) else np.array(pic)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            img = torch.from_numpy(np.ascontiguousarray(arr.transpose((2, 0, 1))))
            return img.to(torch.float32).div(255) if img.dtype == torch.uint8 else img

    class _Unused:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            raise RuntimeError("torchvision stand-in: this transform is not restated (the capture never needs it)")

    tr = _mod("torchvision.transforms", Compose=_Compose, ToTensor=_ToTensor, Normalize=_Unused, Resize=_Unused)
    ops = _mod("torchvision.ops", roi_align=thirdparty.roi_align, roi_pool=None)
    models = _mod("torchvision.models", vgg19=_vgg19)
    _mod("torchvision", transforms=tr, ops=ops, models=models)
    data = _mod("torch_geometric.data", Data=_Data, Batch=_Data)
    gnn = _mod("torch_geometric.nn", GATv2Conv=_GATv2Conv, MessagePassing=_MessagePassing, Sequential=None)
    _mod("torch_geometric", data=data, nn=gnn)

    # FlowNet2's three CUDA extensions (top-level packages `resample2d`, `channelnorm`, `correlation`, built by the
    # reference's setup.py files): module objects whose layers call OUR C restatements (thirdparty.resample2d / ...).
    class _Resample2d(nn.Module):
        def __init__(self, kernel_size=1, bilinear=True):
            super().__init__()
            assert kernel_size == 1 and bilinear

        def forward(self, input1, input2):
            return thirdparty.resample2d(input1, input2)

    class _ChannelNorm(nn.Module):
        def __init__(self, norm_deg=2):
            super().__init__()
            assert norm_deg == 2

        def forward(self, x):
            return thirdparty.channelnorm(x)

    class _Correlation(nn.Module):
        def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
            super().__init__()
            self.cfg = (pad_size, kernel_size, max_displacement, stride1, stride2)

        def forward(self, a, b):
            return thirdparty.correlation(a, b, *self.cfg)

    _mod("resample2d", Resample2d=_Resample2d)
    _mod("channelnorm", ChannelNorm=_ChannelNorm)
    _mod("correlation", Correlation=_Correlation)

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import utils.ops as ref_ops  # noqa: the reference's utils package
    import utils as ref_utils
    from modules.discriminator import discriminator as ref_disc

    def get_grid_cpu(batchsize, rows, cols, gpu_id=0):
        g = torch.zeros([batchsize, 2, rows, cols])
        lx = torch.linspace(-1, 1, cols) if cols > 1 else torch.Tensor([-1])
        g[:, 0] = torch.ger(torch.ones(rows), lx).expand_as(g[:, 0])
        ly = torch.linspace(-1, 1, rows) if rows > 1 else torch.Tensor([-1])
        g[:, 1] = torch.ger(ly, torch.ones(cols)).expand_as(g[:, 1])
        return g

    ref_ops.get_grid = get_grid_cpu
    ref_utils.get_grid = get_grid_cpu

    def get_target_tensor_cpu(self, input_tensor, target_is_real):
        return torch.full_like(input_tensor, self.real_label if target_is_real else self.fake_label)

    ref_disc.GANLoss.get_target_tensor = get_target_tensor_cpu
    return ref_utils


class _Anything(types.ModuleType):
    """Import-only stand-in for a module the reference imports but never calls on the captured path (dominate,
    tensorboard ...): any attribute is another stand-in, calling it returns None."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Anything(self.__name__ + "." + k)
        sys.modules[m.__name__] = m
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


def import_reference(name, max_stubs=32):
    """importlib.import_module(name) for a reference module, injecting an empty stand-in for every absent third-party
    package it trips over (observability / metrics imports of trainer/base.py: dominate, tensorboard ...)."""
    import importlib
    for _ in range(max_stubs):
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            if not e.name or e.name.split(".")[0] in ("modules", "utils", "trainer", "datasets", "losses"):
                raise
            sys.modules[e.name] = _Anything(e.name)
    raise ImportError(name)
