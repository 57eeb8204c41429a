"""CPU-only checks of the host side: C-ABI library, module/state_dict surface, constructor parity, fail-loud."""
import ctypes
import os

import numpy as np
import pytest
import torch

from c2m_amd import _lib, build as c2m_build
from c2m_amd.config import normalize_config, default_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch
from oracle.golden_util import summarize
from golden_io import Case, names


@pytest.fixture(scope="module")
def library():
    if not os.path.exists(_lib.LIB_PATH):
        c2m_build.build()
    return _lib.lib()


def test_abi_exports_every_declared_symbol(library):
    declared = _lib.declared_symbols()
    assert len(declared) >= 24
    missing = [s for s in declared if not hasattr(library, s)]
    assert not missing, f"declared in include/c2m_hip.h but not exported: {missing}"
    assert set(declared) == set(_lib._SIGS), "ctypes signature table out of sync with the header"


def test_abi_host_side_queries(library):
    # pure host helpers: callable without a GPU
    assert library.c2m_conv_wgrad_splits(64, 577, 40 * 128 * 256) >= 1
    assert library.c2m_norm_workspace_floats(2, 3, 100000) == 2 * 3 * 13 * 4
    assert library.c2m_occlusion_splat_workspace_bytes(2, 4, 8) == 2 * 32 * 4 * 3 + 2 * 32 * 4 * 4 * 2
    assert library.c2m_flow_warp_bwd_workspace_bytes(40, 512, 4, 8, 1, 1) >= 40 * 32 * 44
    # Winograd regions per image: 8 x 16 outputs where that fits, another tile shape where it covers the domain with fewer
    assert library.c2m_wino_regions(16, 32) == 4 and library.c2m_wino_regions(128, 256) == 256
    assert library.c2m_wino_regions(18, 34) == 6 and library.c2m_wino_regions(10, 18) == 2
    for h, w in [(1, 1), (9, 11), (34, 66), (66, 130), (130, 258)]:
        assert 1 <= library.c2m_wino_regions(h, w) <= -(-h // 8) * -(-w // 16)


@pytest.mark.parametrize("name", names("e2e_"))
def test_state_dict_surface_and_default_init_match_reference(name):
    c = Case(name)
    cfg = normalize_config(c.meta["cfg"])
    torch.manual_seed(1234 + c.meta["seed"])
    m = GeneratorFullModel(train_params=cfg["train_params"], model_params=cfg["model_params"], dataset="cityscapes")
    sd = m.state_dict()
    ours = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    assert ours == c.meta["spec"], "state_dict keys / order / shapes / dtypes differ from the reference"
    ref = c.group("sum.init")
    for k, v in sd.items():
        np.testing.assert_allclose(summarize(v.float()), ref[k].numpy(), rtol=0, atol=0, err_msg=f"default init of {k}")


def test_product_path_fails_loudly_without_a_gpu():
    cfg = normalize_config(default_config(num_input_frames=1, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16,
                                          out_channel=16, ndf=4))
    m = GeneratorFullModel(train_params=cfg["train_params"], model_params=cfg["model_params"], dataset="cityscapes")
    batch = make_batch(1, 128, 256, 1, seed=0)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(batch)


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import c2m_amd.modules.model, c2m_amd.ops, c2m_amd.losses.losses, c2m_amd.ddp, c2m_amd.train; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, "-c", code], cwd=root)


def test_checkpoint_layout_round_trip_with_torch_adam_states(tmp_path):
    """SURVEY §8f-2 / trainer.py:117-136,245-260: a checkpoint in the reference's layout -- reference state_dict keys (the
    fixture's key spec) + plain torch.optim.Adam state dicts -- loads into the product model, and what the product saves
    has the same layout and loads back (host logic only: no kernel runs)."""
    import copy
    from c2m_amd import checkpoint as ck
    from oracle.golden_util import synth_state
    c = Case("e2e_tin2_spade_full")
    cfg = normalize_config(c.meta["cfg"])
    mk = lambda: GeneratorFullModel(train_params=copy.deepcopy(cfg["train_params"]),
                                    model_params=copy.deepcopy(cfg["model_params"]), dataset="cityscapes")
    model = mk()
    ref_sd = synth_state(c.meta["spec"], 77)                       # what the reference's c2m.state_dict() holds
    # optimizer states as plain torch Adam writes them (the reference's optimizers), over the same parameter lists
    ref_opt = {}
    for key, opt in (("optimizer", model.optimizer), ("optimizer_gnn", model.optimizer_gnn),
                     ("optimizer_d_image", model.d_optimizer_image), ("optimizer_d_video", model.d_optimizer_video)):
        ps = [torch.nn.Parameter(p.detach().clone()) for p in opt.param_groups[0]["params"]]
        t = torch.optim.Adam(ps, lr=opt.param_groups[0]["lr"], betas=opt.param_groups[0]["betas"],
                             eps=opt.param_groups[0]["eps"])
        for i, p in enumerate(ps[:5]):                             # a few parameters have state
            p.grad = torch.full_like(p, 0.01 * (i + 1))
        t.step()
        ref_opt[key] = t.state_dict()
    d = tmp_path / "samples"
    d.mkdir()
    torch.save(dict(c2m=ref_sd, **ref_opt), str(d / "latest_c2m_model.pth.tar"))
    np.savetxt(str(d / "iter.txt"), (4, 120), delimiter=",", fmt="%d")
    assert ck.load_checkpoint(model, str(d)) == (4, 120)
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in ref_sd.items():
        assert torch.equal(sd[k].cpu(), v), k
    st = model.optimizer.state_dict()["state"]
    assert sorted(st.keys()) == [0, 1, 2, 3, 4] and float(st[0]["step"]) == 1.0
    torch.testing.assert_close(st[2]["exp_avg"], ref_opt["optimizer"]["state"][2]["exp_avg"])
    assert model.optimizer_gnn.state_dict()["state"] == {}, "the reference does not restore optimizer_gnn (trainer.py:124-129)"
    # save from the product, load into a fresh product model and into plain torch optimizers
    path = ck.save_checkpoint(model, str(d), current_epoch=7, epoch_iter=33)
    saved = torch.load(path, weights_only=False)
    assert set(saved) == {"c2m", "optimizer", "optimizer_gnn", "optimizer_d_image", "optimizer_d_video"}
    assert list(saved["c2m"].keys()) == list(ref_sd.keys())
    other = mk()
    assert ck.load_checkpoint(other, str(d)) == (8, 33)
    for k, v in other.state_dict().items():
        assert torch.equal(v, sd[k]), k
    ps = [torch.nn.Parameter(p.detach().clone()) for p in model.optimizer.param_groups[0]["params"]]
    t = torch.optim.Adam(ps, lr=1e-4, betas=(0.5, 0.999), eps=1e-7)
    t.load_state_dict(saved["optimizer"])                          # our optimizer state is a valid torch Adam state
    assert float(t.state_dict()["state"][0]["step"]) == 1.0


def test_stream_batch_windows_stack_along_the_batch_axis():
    """BASELINE configs[4]: a 14-frame stream sample = two 7-frame windows; the per-rank batch doubles."""
    from c2m_amd.synthetic import make_batch, make_stream_batch
    b = make_stream_batch(streams=2, windows=2, height=32, width=64, num_objects=[3, 2, 3, 2])
    assert b["video"].shape == (4, 3, 7, 32, 64) and b["target_bw_of"].shape == (4, 2, 5, 32, 64)
    assert b["tracking_gnn"].num_real_nodes.tolist() == [3, 2, 3, 2] and b["tracking_gnn"].num_nodes == 10
    ref = make_batch(4, 32, 64, num_objects=[3, 2, 3, 2])
    assert torch.equal(b["video"], ref["video"]) and torch.equal(b["instance_mask"], ref["instance_mask"])
    with pytest.raises(ValueError):
        make_stream_batch(streams=1, windows=0)


def test_gatv2_dense_product_form_vs_edge_list_oracle_form():
    """torch_geometric's GATv2Conv is an absent third-party dependency (parity unpinned).  The product
    computes it in DENSE form (all ordered pairs, edge-multiplicity mask), the oracle over the EDGE LIST (scatter /
    index_add) -- two independent restatements of the published layer that must agree, including duplicate edges, nodes
    without incoming edges and gradients.  Above DENSE_MAX_NODES the product switches to padded rows of edges per target
    node (dense reductions along the row: no float-atomic index_add) -- a third formulation, held to the oracle's
    here and to the product's own dense form in test_gatv2_padded_rows_form_equals_the_dense_form."""
    import torch
    from c2m_amd.thirdparty import GATv2Conv
    from oracle import thirdparty as TP
    torch.manual_seed(0)
    m = GATv2Conv(16, 8, heads=4, concat=False, add_self_loops=False)
    ei = torch.tensor([[0, 1, 2, 0, 2, 1, 4, 5, 5], [1, 0, 0, 2, 1, 2, 5, 4, 4]])        # 5 -> 4 twice; nodes 3, 6 isolated
    for n in (7, 70):                                                                    # dense form / edge-list form
        x = torch.randn(n, 16)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = m(xa, ei)
        yr = TP.gatv2_conv(xb, ei, m.lin_l.weight, m.lin_l.bias, m.lin_r.weight, m.lin_r.bias, m.att, m.bias, 4)
        torch.testing.assert_close(y, yr, rtol=1e-5, atol=1e-6)
        assert torch.equal(y[3], m.bias) and torch.equal(y[6], m.bias)                   # no incoming edge: bias only
        go = torch.randn_like(y)
        (y * go).sum().backward()
        (yr * go).sum().backward()
        torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-6)


def test_gatv2_padded_rows_form_equals_the_dense_form():
    """The > 64-node branch of the product's GATv2Conv against its dense branch on the SAME graph (90 nodes, ragged in-degrees,
    duplicate edges, isolated nodes): forward, input gradient and every parameter gradient; twice the same bits."""
    import torch
    from c2m_amd.thirdparty import GATv2Conv
    torch.manual_seed(1)
    n = 90
    g = torch.Generator().manual_seed(3)
    ei = torch.randint(0, n - 5, (2, 400), generator=g)                # nodes 85 .. 89 isolated; duplicates certain
    ei = torch.cat([ei, ei[:, :7]], 1)
    x = torch.randn(n, 16, generator=g)
    outs = []
    for dense_max in (64, 1000, 64):
        m = GATv2Conv(16, 8, heads=4, concat=False, add_self_loops=False)
        m.load_state_dict(outs[0][3]) if outs else None
        m.DENSE_MAX_NODES = dense_max
        xa = x.clone().requires_grad_(True)
        y = m(xa, ei)
        (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
        outs.append((y.detach(), xa.grad, [p.grad.clone() for p in m.parameters()], m.state_dict()))
    (y0, gx0, gp0, _), (y1, gx1, gp1, _), (y2, gx2, gp2, _) = outs
    torch.testing.assert_close(y0, y1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gx0, gx1, rtol=1e-4, atol=1e-6)
    for a, b in zip(gp0, gp1):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    assert torch.equal(y0[-1], m.bias.detach())
    assert torch.equal(y0, y2) and torch.equal(gx0, gx2) and all(torch.equal(a, b) for a, b in zip(gp0, gp2))


@pytest.mark.parametrize("T", [2, 3, 4, 5, 7])
def test_time_pair_table_is_the_adjoint_of_reflect_pad_plus_three_taps(T):
    """ops._time_pair_table (temporal reflect padding folded into the 3x3x3 Winograd data gradient, c2m_conv_wino geom[33]):
    summing dY frame `to` through tap kt over a frame's pairs must equal autograd through F.pad(reflect) + a 3-tap correlation
    in time (src/modules/layers/common.py Conv3d call sites use padding_mode='reflect')."""
    import torch.nn.functional as F
    from c2m_amd.ops import _time_pair_table
    g = torch.Generator().manual_seed(T)
    x = torch.randn(1, 1, T, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(1, 1, 3, generator=g, dtype=torch.float64)
    gy = torch.randn(1, 1, T, generator=g, dtype=torch.float64)
    y = F.conv1d(F.pad(x, (1, 1), mode="reflect"), w)
    y.backward(gy)
    tab = _time_pair_table(T)
    got = torch.zeros(T, dtype=torch.float64)
    for t in range(T):
        n = int(tab[t, 0])
        assert 2 <= n <= 5 and (tab[t, 1 + 2 * n:] == 0).all()
        for j in range(n):
            to, ublock = int(tab[t, 1 + 2 * j]), int(tab[t, 2 + 2 * j])
            got[t] += gy[0, 0, to] * w[0, 0, 2 - ublock]          # U block = flipped time tap
    assert torch.allclose(got, x.grad[0, 0], rtol=0, atol=1e-14)


def test_conv_plan_rejects_a_channel_mismatch():
    """torch.nn.functional.conv2d raises on an input whose channel count is not the weight's; so does the plan (the pack kernels
    would otherwise read past the end of the weight tensor)."""
    import torch
    from c2m_amd import ops
    with pytest.raises(RuntimeError, match="channels"):
        ops._ConvPlan((2, 32, 16, 32), (16, 16, 3, 3), (1, 1, 1), (0, 1, 1), False, torch.device("cpu"))


def test_geom_names_one_definition_for_kernels_and_host():
    """the geom[] blocks of the convolution entry points are addressed by NAME -- include/c2m_geom.h is compiled
    into the library (csrc/common.h) and parsed by the ctypes host; the loaded library must report the header's ABI version and
    block lengths, no two names may share an index, and the per-class ranges may not run into the flag entries (the round-3 bug:
    class 6's pad offsets on geom[90..92]; now also a static_assert of the header)."""
    from c2m_amd import _lib, ops
    L = _lib.lib()
    G, WG = _lib.GEOM, _lib.WINO_GEOM
    assert (L.c2m_abi_version(), L.c2m_geom_len(), L.c2m_wino_geom_len()) == (_lib.ABI_VERSION, G.LEN, WG.LEN) == (6, 120, 36)
    assert len(set(G.values())) == len(G) and len(set(WG.values())) == len(WG)
    assert G.CLS_OUT_OFF + 8 <= G.X_TYPE and G.G8_VARIANT < G.CLS_PO and G.CLS_PO + 24 == G.LEN
    assert G.PATCH_TY + 3 == G.PATCH_TX and G.PS_T + 16 == G.PATCH        # the two-target block is 16 consecutive entries
    with pytest.raises(AttributeError):
        G.NO_SUCH_ENTRY
    # the host builds its blocks with these names: a class-batched plan (8 stride-parity classes of a 4x4x4 stride-2 layer)
    import torch
    pl = ops._ConvPlan((2, 16, 4, 16, 32), (32, 16, 4, 4, 4), (2, 2, 2), (1, 1, 1), True, torch.device("cpu"))
    g = pl.cls_batch["groups"][0]["geom"]
    assert len(g) == G.LEN and int(g[G.NCLS]) == 8 and int(g[G.X_TYPE]) == 0 and int(g[G.Y_TYPE]) == 0 and int(g[G.WGRAD_WIDE]) == 0
    assert [tuple(int(v) for v in g[G.CLS_PO + 3 * c:G.CLS_PO + 3 * c + 3]) for c in range(8)] == \
        [(a, b, c) for a in range(2) for b in range(2) for c in range(2)]
