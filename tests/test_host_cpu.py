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
    assert library.c2m_flow_warp_bwd_needs_zero(40, 512, 4, 8) in (0, 1)


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
