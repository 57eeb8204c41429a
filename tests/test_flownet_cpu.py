"""SURVEY 8f-4 on CPU: the FlowNet2 parameter surface and initialisation of the product equal the live reference's
(tests/golden/flownet2_compute_flow.npz, captured from flownet2/models.py by oracle/capture_golden.py::capture_flownet), and the
C restatements of its three custom operators satisfy the properties their definitions imply (they are "parity unpinned":
the reference's CUDA extensions cannot be built here and hold no fixtures)."""
import types

import numpy as np
import torch
import torch.nn.functional as F

from oracle import thirdparty as TP
from oracle.golden_util import summarize
from golden_io import Case


def test_flownet2_state_dict_surface_and_seeded_init_match_the_reference():
    from c2m_amd.modules.third_party.flow_net import flownet2
    c = Case("flownet2_compute_flow")
    torch.manual_seed(c.meta["seed"])
    net = flownet2.FlowNet2(types.SimpleNamespace(fp16=False, rgb_max=1.0))
    sd = net.state_dict()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == c.meta["spec"]     # keys, order, shapes
    assert sum(p.numel() for p in net.parameters()) == c.meta["nparams"] == 162518834      # models.py:17
    got = np.stack([summarize(v) for v in sd.values()])
    assert np.array_equal(got, c.arr["init.fingerprints"]), "same seed must give the reference's initial weights, key by key"


def test_flownet_wrapper_refuses_missing_pretrained_weights(tmp_path, monkeypatch):
    from c2m_amd.modules.third_party.flow_net.flow_net import FlowNet
    import pytest
    monkeypatch.setenv("C2M_FLOWNET2_CHECKPOINT", str(tmp_path / "absent.pth.tar"))
    with pytest.raises(FileNotFoundError):
        FlowNet(pretrained=True)


def test_oracle_resample2d_properties():
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 9, 13, generator=g)
    assert torch.equal(TP.resample2d(img, torch.zeros(2, 2, 9, 13)), img)                       # zero flow: identity
    shift = torch.zeros(2, 2, 9, 13)
    shift[:, 0] = 2.0                                                                             # x + 2, clamped at the border
    want = torch.cat([img[..., 2:], img[..., -1:].expand(-1, -1, -1, 2)], -1)
    assert torch.equal(TP.resample2d(img, shift), want)
    flow = 3.0 * torch.randn(2, 2, 9, 13, generator=g)
    # against grid_sample(bilinear, border, align_corners=True) on the same pixel coordinates (the ops differ only in fp order)
    ys, xs = torch.meshgrid(torch.arange(9.0), torch.arange(13.0), indexing="ij")
    gx = ((xs + flow[:, 0]).clamp(0, 12) / 12) * 2 - 1
    gy = ((ys + flow[:, 1]).clamp(0, 8) / 8) * 2 - 1
    ref = F.grid_sample(img, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border", align_corners=True)
    inside = ((xs + flow[:, 0] >= 0) & (xs + flow[:, 0] <= 12) & (ys + flow[:, 1] >= 0) & (ys + flow[:, 1] <= 8)).unsqueeze(1)
    assert ((TP.resample2d(img, flow) - ref).abs() * inside).max() < 1e-5


def test_oracle_channelnorm_and_correlation_properties():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 6, 7, generator=g)
    assert torch.allclose(TP.channelnorm(x), x.pow(2).sum(1, keepdim=True).sqrt(), rtol=1e-6, atol=1e-7)
    a, b = torch.randn(1, 8, 10, 12, generator=g), torch.randn(1, 8, 10, 12, generator=g)
    out = TP.correlation(a, b, pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=2)
    assert out.shape == (1, 25, 10, 12)
    # displacement (tj, ti) in units of stride2: out[(tj+2)*5 + (ti+2)] = mean_c a[y, x] * b[y + 2 tj, x + 2 ti] (zero outside)
    bp = F.pad(b, (4, 4, 4, 4))
    for tj in range(-2, 3):
        for ti in range(-2, 3):
            want = (a * bp[:, :, 4 + 2 * tj:14 + 2 * tj, 4 + 2 * ti:16 + 2 * ti]).sum(1) / 8
            assert torch.allclose(out[:, (tj + 2) * 5 + (ti + 2)], want, rtol=1e-5, atol=1e-6), (tj, ti)
    assert torch.allclose(out[:, 12], (a * b).mean(1), rtol=1e-5, atol=1e-6)                      # zero displacement
    # 3x3 patch, stride1 = 2 (general form of correlation_cuda_kernel.cu): size and centre value
    out3 = TP.correlation(a, b, pad_size=3, kernel_size=3, max_displacement=2, stride1=2, stride2=1)
    assert out3.shape == (1, 25, 5, 6)
    ap, bp3 = F.pad(a, (3, 3, 3, 3)), F.pad(b, (3, 3, 3, 3))
    y1, x1 = 2 * 1 + 2, 2 * 2 + 2                                                                 # output (1, 2) in padded coordinates
    want = sum((ap[0, :, y1 + j, x1 + i] * bp3[0, :, y1 + j, x1 + i]).sum() for j in (-1, 0, 1) for i in (-1, 0, 1)) / (9 * 8)
    assert abs(float(out3[0, 12, 1, 2]) - float(want)) < 1e-5
