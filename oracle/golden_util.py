"""TEST INFRASTRUCTURE -- helpers shared by oracle/capture_golden.py and tests/.

Golden fixtures must stay small, so instead of storing network weights we store the *state_dict
surface* (key, shape, dtype) captured from the reference and regenerate the values from a seed with
`synth_state`, a reference-independent deterministic initialiser keyed on the parameter name.
The capture script loads exactly these values into the reference model before running it.
"""
import zlib

import numpy as np
import torch

VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def synth_tensor(key, shape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("vgg19.mean"):
        return torch.tensor(VGG_MEAN).view(shape)
    if key.endswith("vgg19.std"):
        return torch.tensor(VGG_STD).view(shape)
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith(("weight_u", "weight_v")):
        v = torch.randn(shape, generator=g)
        return v / v.norm()
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return torch.randn(shape, generator=g) * (1.0 / fan_in ** 0.5)
    if key.endswith("weight"):  # 1-D: norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)


def synth_state(spec, seed):
    """spec: iterable of (key, shape, dtype-string)."""
    return {k: synth_tensor(k, s, d, seed) for k, s, d in spec}


def state_spec(state_dict):
    return [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()]


def summarize(t):
    """Compact, order-insensitive-ish fingerprint of a float tensor: [sum, abs-sum, sq-sum, first, last] in float64."""
    t = t.detach().double().reshape(-1)
    if t.numel() == 0:
        return np.zeros(5)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item(), t[0].item(), t[-1].item()])


def pack_mask(t):
    """Bit-pack a {0,1} float tensor (index/mask paths are compared bit for bit)."""
    a = t.detach().cpu().numpy()
    assert np.all((a == 0) | (a == 1)), "not a binary mask"
    return np.packbits(a.astype(np.uint8).reshape(-1)), np.array(a.shape)


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(np.float32)


# ---- compact module fixtures (round 2): inputs regenerated from seeds, large tensors stored as fingerprint + subsample
COMPACT_LIMIT = 4096


def synth_input(spec):
    """spec = dict(seed, shape, kind in {"randn", "rand", "mask"}, scale): a seeded CPU tensor (torch's CPU generator is
    deterministic across hosts), so fixtures need not store the inputs."""
    g = torch.Generator(device="cpu").manual_seed(int(spec["seed"]))
    shape = tuple(spec["shape"])
    if spec["kind"] == "randn":
        return torch.randn(shape, generator=g) * float(spec.get("scale", 1.0))
    if spec["kind"] == "rand":
        return torch.rand(shape, generator=g) * float(spec.get("scale", 1.0))
    if spec["kind"] == "mask":
        return (torch.rand(shape, generator=g) > float(spec.get("scale", 0.3))).float()
    raise KeyError(spec["kind"])


def compact(prefix, name, t):
    """{key: array} for one tensor: whole if small, else [sum, abs-sum, sq-sum, first, last] + every k-th element."""
    t = t.detach().cpu()
    if t.numel() <= COMPACT_LIMIT:
        return {f"{prefix}.{name}": t.numpy()}
    k = -(-t.numel() // COMPACT_LIMIT)
    return {f"sum{prefix}.{name}": summarize(t.float()), f"sub{prefix}.{name}": t.reshape(-1)[::k].clone().numpy()}


def check_compact(arr, prefix, name, got, tol, what, floor=0.0):
    """Compare `got` with the stored form of <prefix>.<name>; max-abs error relative to the tensor's scale."""
    got = got.detach().cpu().double()
    key = f"{prefix}.{name}"
    if key in arr:
        ref = torch.from_numpy(np.array(arr[key])).double()
        assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
        scale = max(float(ref.abs().max()), floor, 1e-30)
        err = float((got - ref).abs().max())
        assert err <= tol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"
        return
    ref_sub = torch.from_numpy(np.array(arr[f"sub{prefix}.{name}"])).double()
    k = -(-got.numel() // COMPACT_LIMIT)
    sub = got.reshape(-1)[::k]
    assert sub.shape == ref_sub.shape, f"{what}: subsample {tuple(sub.shape)} vs {tuple(ref_sub.shape)}"
    scale = max(float(ref_sub.abs().max()), floor, 1e-30)
    err = float((sub - ref_sub).abs().max())
    assert err <= tol * scale, f"{what}: subsample max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"
    rs = np.array(arr[f"sum{prefix}.{name}"])
    gs = summarize(got)
    n = got.numel()
    assert abs(gs[1] - rs[1]) <= tol * abs(rs[1]) + tol * floor * n, f"{what}: abs-sum {gs[1]} vs {rs[1]}"
    assert abs(gs[2] - rs[2]) <= 2 * tol * abs(rs[2]) + (tol * floor) ** 2 * n, f"{what}: sq-sum {gs[2]} vs {rs[2]}"
