"""SURVEY 8f-4 on the GPU: FlowNet2's three custom operators (flownet_ops.hip) bit-exact against the C restatement, and the
whole online target-flow path -- FlowNet2 forward on the HIP conv kernels, FlowNet.compute_flow_and_conf, compute_flow --
against what the LIVE reference's Python produced (tests/golden/flownet2_compute_flow.npz; the reference's module graph, init
and Trainer.compute_flow ran on CPU over the same C restatements of its CUDA extensions)."""
import types

import numpy as np
import pytest
import torch

from c2m_amd import ops
from c2m_amd.train import compute_flow
from oracle import thirdparty as TP
from oracle.golden_util import check_compact, synth_input
from golden_io import Case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _r(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("shape", [(2, 3, 64, 128), (1, 5, 9, 13), (3, 1, 17, 33)])
def test_resample2d_bitexact_vs_c_oracle(shape):
    N, C, H, W = shape
    img = _r(1, *shape)
    flow = _r(2, N, 2, H, W, scale=4.0)
    flow[0, :, 0, :] += 500.0               # far outside: all four taps clamp to the border
    flow[-1, :, :, 0] -= 500.0
    flow[0, :, 1, :] = torch.round(flow[0, :, 1, :])      # integer displacements: alpha = beta = 0
    got = ops.resample2d(img.to(DEV), flow.to(DEV)).cpu()
    assert torch.equal(got, TP.resample2d(img, flow))


@pytest.mark.parametrize("shape", [(2, 3, 64, 128), (1, 7, 5, 9), (2, 2, 128, 256)])
def test_channelnorm_bitexact_vs_c_oracle(shape):
    x = _r(3, *shape)
    assert torch.equal(ops.channelnorm(x.to(DEV)).cpu(), TP.channelnorm(x))


@pytest.mark.parametrize("cfg", [((1, 256, 8, 16), dict(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2)),   # FlowNetC
                                 ((2, 8, 10, 12), dict(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=2)),
                                 ((1, 6, 11, 9), dict(pad_size=3, kernel_size=3, max_displacement=2, stride1=2, stride2=1)),
                                 ((1, 4, 7, 7), dict(pad_size=0, kernel_size=1, max_displacement=2, stride1=1, stride2=1))])
def test_correlation_bitexact_vs_c_oracle(cfg):
    shape, kw = cfg
    a, b = _r(4, *shape), _r(5, *shape)
    got = ops.correlation(a.to(DEV), b.to(DEV), **kw).cpu()
    want = TP.correlation(a, b, **kw)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("case", [((2, 6, 8, 12), 10, True), ((1, 2, 16, 32), 2, False), ((3, 130, 4, 4), 64, True)])
def test_conv_transpose2d_vs_torch(case):
    """ConvTranspose2d(4, 2, 1) (+ bias + LeakyReLU 0.1) on the stride-parity data-gradient kernels."""
    xs, cout, bias = case
    x = _r(6, *xs)
    w = _r(7, xs[1], cout, 4, 4, scale=(1.0 / (xs[1] * 4)) ** 0.5)
    b = _r(8, cout, scale=0.3) if bias else None
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=1), 0.1)
    with torch.no_grad():
        got = ops.conv_transpose2d(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), 2, 1, act="lrelu", slope=0.1).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    with pytest.raises(RuntimeError):
        ops.conv_transpose2d(x.to(DEV).requires_grad_(True), w.to(DEV))


@pytest.fixture(scope="module")
def flownet():
    from c2m_amd.modules.third_party.flow_net.flow_net import FlowNet
    c = Case("flownet2_compute_flow")
    torch.manual_seed(c.meta["seed"])                # the reference's seeded init (checked key by key in test_flownet_cpu.py)
    return c, FlowNet(pretrained=False).to(DEV)


def test_flownet2_pair_vs_reference_capture(flownet):
    c, net = flownet
    video = synth_input(c.meta["video"]).to(DEV)
    a, b = video[:, :, 1] * 2 - 1, video[:, :, 2] * 2 - 1
    flow, conf = net(a, b)
    assert flow.shape == (1, 2, 64, 128) and conf.shape == (1, 1, 64, 128) and not flow.requires_grad
    with torch.no_grad():
        netc = net.flowNet.flownetc(torch.cat((a - a.mean(), b - b.mean()), 1))[0]
    check_compact(c.arr, "pair", "netc", netc, 1e-3, "FlowNetC flow2 (conv stack + correlation + refinement decoder)")
    check_compact(c.arr, "pair", "flow", flow, 2e-3, "FlowNet2 flow")
    check_compact(c.arr, "pair", "conf", conf, 5e-3, "occlusion map of the flow")
    # 5-D input form (flow_net.py:38-50)
    f5, c5 = net(a.unsqueeze(1), b.unsqueeze(1))
    assert f5.shape == (1, 1, 2, 64, 128) and torch.equal(f5[:, 0], flow) and torch.equal(c5[:, 0], conf)


def test_flownet_resizes_to_multiples_of_64(flownet):
    c, net = flownet
    video = synth_input(c.meta["video"])
    a, b = video[:, :, 1] * 2 - 1, video[:, :, 2] * 2 - 1
    odd_a = torch.nn.functional.interpolate(a, size=(80, 144), mode="bilinear", align_corners=False).to(DEV)
    odd_b = torch.nn.functional.interpolate(b, size=(80, 144), mode="bilinear", align_corners=False).to(DEV)
    flow, conf = net(odd_a, odd_b)
    assert flow.shape == (1, 2, 80, 144)
    check_compact(c.arr, "odd", "flow", flow, 2e-3, "flow at 80x144 (computed at 64x128, resized back, scaled by 80/64)")
    check_compact(c.arr, "odd", "conf", conf, 5e-3, "confidence at 80x144")


def test_compute_flow_vs_reference_trainer(flownet):
    """Trainer.compute_flow (trainer.py:42-98) of the live reference vs c2m_amd.train.compute_flow (one batched pass)."""
    c, net = flownet
    video = synth_input(c.meta["video"]).to(DEV)
    out = compute_flow(net, {"video": video}, dict(num_input_frames=2, num_predicted_frames=5, use_fw_of=True))
    assert out["input_of"].shape == (1, 2, 1, 64, 128) and out["target_bw_of"].shape == (1, 2, 5, 64, 128)
    assert out["target_bw_occ"].shape == (1, 1, 5, 64, 128)
    for k in ("input_of", "target_bw_of", "target_fw_of"):
        check_compact(c.arr, "cf", k, out[k], 2e-3, k)
    for k in ("input_occ", "target_bw_occ", "target_fw_occ"):
        check_compact(c.arr, "cf", k, out[k], 5e-3, k)
    out1 = compute_flow(net, {"video": video}, dict(num_input_frames=1, num_predicted_frames=5))
    assert out1["input_of"] is None and out1["input_occ"] is None and "target_fw_of" not in out1
    # t_in = 1 on the clip shifted by one frame asks for the same (frame 1 -> frame 2 + i) pairs as t_in = 2 on the whole clip
    sh = compute_flow(net, {"video": video[:, :, 1:].contiguous()}, dict(num_input_frames=1, num_predicted_frames=5))
    assert sh["input_of"] is None
    torch.testing.assert_close(sh["target_bw_of"], out["target_bw_of"], rtol=1e-4, atol=1e-4)
    assert float((sh["target_bw_occ"] - out["target_bw_occ"]).abs().mean()) < 1e-4
    # ... and on the unshifted clip the targets are different pairs (frame 0 -> frame 1 + i)
    assert not torch.allclose(out1["target_bw_of"], out["target_bw_of"], atol=1e-3)


def test_flownet_is_fp32_inside_a_bf16_training_mode(flownet):
    """under ops.set_conv_precision('bf16') (configs[2-4]) the flow net's convolutions returned bf16 tensors that
    the fp32-only correlation / resample2d / channelnorm kernels then read as floats.  The net now pins fp32 for itself and
    the three operators cast; targets are bit-identical in both modes."""
    from c2m_amd import ops
    c, net = flownet
    video = synth_input(c.meta["video"]).to(DEV)
    tp = dict(num_input_frames=2, num_predicted_frames=5)
    ref = compute_flow(net, {"video": video}, tp)
    with ops.conv_precision("bf16"):
        got = compute_flow(net, {"video": video}, tp)
        a = torch.randn(2, 16, 24, 32, device=DEV)
        b = torch.randn(2, 16, 24, 32, device=DEV)
        fl = torch.randn(2, 2, 24, 32, device=DEV)
        ab, bb = a.bfloat16(), b.bfloat16()
        assert torch.equal(ops.correlation(ab, bb), ops.correlation(ab.float(), bb.float()))
        assert torch.equal(ops.channelnorm(ab), ops.channelnorm(ab.float()))
        assert torch.equal(ops.resample2d(ab, fl), ops.resample2d(ab.float(), fl))
    for k in ("input_of", "input_occ", "target_bw_of", "target_bw_occ"):
        assert torch.equal(got[k], ref[k]), k


def test_online_flow_feeds_the_training_step(flownet):
    """use_pre_processed_of False end to end: targets from the flow net drive one generator step (trainer.py:109-110)."""
    import copy
    from c2m_amd.config import default_config, normalize_config
    from c2m_amd.modules.model import GeneratorFullModel
    from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
    from c2m_amd.train import TrainStep
    _, net = flownet
    cfg = normalize_config(default_config(num_input_frames=2, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16,
                                          out_channel=16, ndf=4, use_spade=True))
    tp = cfg["train_params"]
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=copy.deepcopy(tp), model_params=copy.deepcopy(cfg["model_params"]),
                               dataset="cityscapes").to(DEV).train()
    batch = batch_to(make_batch(1, 128, 256, 2, seed=3), DEV)
    batch.update(compute_flow(net, batch, tp))
    rng = make_step_rng(batch, z_dim=16, latent_dim=32, seed=0)
    batch["rng"] = {k: v.to(DEV) for k, v in rng.items()}
    _, lg, _ = TrainStep(model, run_optimizers=True, distributed=False)(batch)
    assert all(np.isfinite(float(v.detach())) for v in lg.values())
