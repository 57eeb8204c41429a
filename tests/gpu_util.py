import torch


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def rel_close(a, b, tol, what="", floor=0.0):
    """Norm-wise relative error (GEMM accumulation order differs between CPU and MFMA).  `floor` is a lower bound for
    the scale: gradients that are analytically zero (a conv bias in front of a BatchNorm) are pure rounding noise."""
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), floor, 1e-30)
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def rnd(seed, *shape, scale=1.0):
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed)) * scale
