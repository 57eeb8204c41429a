"""ctypes binding of libc2m_hip.so (include/c2m_hip.h).  There is NO fallback: if the library is missing or a tensor
is not on a HIP device, the op raises -- the product path never silently runs anything else."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C2M_AMD_LIB") or os.path.join(_HERE, "lib", "libc2m_hip.so")   # override: tuning builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "c2m_hip.h")
_lib = None

c_void_p, c_int, c_long, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double

_SIGS = {
    "c2m_abi_version": (c_int, []),
    "c2m_geom_len": (c_int, []),
    "c2m_wino_geom_len": (c_int, []),
    "c2m_conv_igemm": (c_int, [c_void_p] * 7 + [c_int, c_float, c_void_p]),
    "c2m_nchw_to_nc8": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_void_p]),
    "c2m_conv_patch_nc8": (c_int, [c_void_p] * 6 + [c_int, c_float, c_void_p]),
    "c2m_conv_s2_nc8": (c_int, [c_void_p] * 4 + [c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "c2m_conv3d_nc8": (c_int, [c_void_p] * 4 + [c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "c2m_conv3d_dgrad_nc8": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "c2m_conv_wgrad3d_nc8": (c_int, [c_void_p] * 5 + [c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "c2m_conv_s2_dgrad_nc8": (c_int, [c_void_p] * 3 + [c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "c2m_pack_weights_bf16_s2_bytes": (c_long, [c_int, c_int]),
    "c2m_conv_wgrad_nc8_splits": (c_int, [c_int, c_int, c_long, c_int, c_int, c_int]),
    "c2m_conv_wgrad_nc8_slab_floats": (c_long, [c_int, c_int, c_long, c_int, c_int, c_int]),
    "c2m_conv_wgrad_nc8": (c_int, [c_void_p] * 5 + [c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "c2m_reflect_border_add": (c_int, [c_void_p, c_void_p, c_long] + [c_int] * 7 + [c_void_p]),
    "c2m_conv_igemm_splits": (c_int, [c_int, c_int, c_int]),
    "c2m_splitk_reduce": (c_int, [c_void_p] * 3 + [c_long, c_int, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    "c2m_conv_wgrad_splits": (c_int, [c_int, c_int, c_int]),
    "c2m_conv_wgrad_rows": (c_int, [c_int, c_int]),
    "c2m_conv_wgrad": (c_int, [c_void_p] * 8),
    "c2m_reflect_fold": (c_int, [c_void_p, c_void_p, c_long] + [c_int] * 7 + [c_void_p]),
    "c2m_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "c2m_pack_job_bytes": (c_int, []),
    "c2m_pack_job_fill": (c_long, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_uint]),
    "c2m_pack_multi": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "c2m_pack_weights_bf16_patch_bytes": (c_long, [c_int, c_int]),
    "c2m_pack_weights_bf16_patch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "c2m_pack_weights_bf16_gather_bytes": (c_long, [c_void_p]),
    "c2m_pack_weights_bf16_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "c2m_wino_upack_floats": (c_long, [c_int, c_int]),
    "c2m_wino_filter_transform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "c2m_conv_wino": (c_int, [c_void_p] * 6 + [c_int, c_float, c_void_p]),
    "c2m_wino_regions": (c_int, [c_int, c_int]),
    "c2m_wino4_upack_floats": (c_long, [c_int, c_int]),
    "c2m_wino4_filter_transform": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "c2m_wino4_regions": (c_int, [c_int, c_int]),
    "c2m_conv_wino4": (c_int, [c_void_p] * 6 + [c_int, c_float, c_void_p]),
    "c2m_ring_pack_floats": (c_long, [c_int, c_int]),
    "c2m_ring_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "c2m_reflect_ring_dgrad": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "c2m_reflect_ring_buffer": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "c2m_wino_wgrad_splits": (c_int, [c_int] * 5),
    "c2m_conv_wino_wgrad": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "c2m_conv_wino_wgrad3d": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "c2m_prep_video": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "c2m_prep_seg_onehot": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "c2m_prep_flow_occ": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "c2m_event_create": (c_int, [c_void_p]),
    "c2m_event_record": (c_int, [c_void_p, c_void_p]),
    "c2m_event_elapsed_ms": (c_int, [c_void_p, c_void_p, c_void_p]),
    "c2m_event_destroy": (c_int, [c_void_p]),
    "c2m_adam_chunk": (c_int, []),
    "c2m_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_double] * 5 + [c_void_p]),
    "c2m_norm_workspace_floats": (c_long, [c_int, c_int, c_long]),
    "c2m_norm_stats": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_int, c_float, c_float, c_int, c_void_p]),
    "c2m_norm_fwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_long, c_int, c_float, c_float, c_int, c_float, c_int, c_void_p]),
    "c2m_norm_set_fused": (c_int, [c_int]),
    "c2m_norm_apply": (c_int, [c_void_p] * 8 + [c_int, c_int, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    "c2m_norm_bwd": (c_int, [c_void_p] * 13 + [c_int, c_int, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    "c2m_act_bwd": (c_int, [c_void_p] * 3 + [c_long, c_int, c_float, c_int, c_void_p]),
    "c2m_grad_to_nc8": (c_int, [c_int] + [c_void_p] * 5 + [c_long, c_int, c_long, c_long, c_int, c_float, c_void_p]),
    "c2m_resample2d_fwd": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "c2m_channelnorm_fwd": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "c2m_correlation_out_size": (c_int, [c_int] * 5),
    "c2m_correlation_fwd": (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    "c2m_bias_act": (c_int, [c_void_p, c_void_p, c_long, c_int, c_long, c_int, c_float, c_void_p]),
    "c2m_flow_warp_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "c2m_flow_warp_bwd_workspace_bytes": (c_long, [c_int] * 6),
    "c2m_flow_warp_bwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p, c_int, c_void_p]),
    "c2m_resize_bilinear": (c_int, [c_void_p, c_void_p, c_long] + [c_int] * 5 + [c_double, c_int, c_void_p]),
    "c2m_upsample2x_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "c2m_upsample2x_nc8": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "c2m_gat_dense_fwd": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_void_p]),
    "c2m_gat_dense_bwd_workspace_floats": (c_long, [c_int, c_int, c_int]),
    "c2m_gat_dense_bwd": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_float, c_void_p]),
    "c2m_upsample2x_bwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "c2m_roi_align_fwd": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_float, c_void_p]),
    "c2m_roi_align_bwd": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_float, c_void_p]),
    "c2m_maxpool2x2_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    "c2m_maxpool2x2_bwd": (c_int, [c_void_p] * 3 + [c_long, c_int, c_int, c_int, c_void_p]),
    "c2m_maxpool2x2_relu_bwd": (c_int, [c_void_p] * 3 + [c_long, c_int, c_int, c_int, c_void_p]),
    "c2m_sparse_raster": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "c2m_occlusion_splat_workspace_bytes": (c_long, [c_long, c_int, c_int]),
    "c2m_occlusion_splat": (c_int, [c_void_p, c_long, c_long, c_long] + [c_int] * 4 + [c_void_p] * 4),
    "c2m_l1_mean_fwd": (c_int, [c_void_p] * 4 + [c_long, c_int, c_long, c_void_p, c_int, c_void_p]),
    "c2m_relu_tap_bwd": (c_int, [c_void_p] * 5 + [c_long, c_int, c_void_p]),
    "c2m_l1_mean_bwd": (c_int, [c_void_p] * 6 + [c_long, c_int, c_long, c_int, c_void_p]),
    "c2m_ssim_fwd": (c_int, [c_void_p] * 3 + [c_long, c_int, c_int, c_void_p, c_void_p]),
    "c2m_ssim_bwd": (c_int, [c_void_p] * 5 + [c_long, c_int, c_int, c_void_p]),
}


GEOM_HEADER = os.path.join(os.path.dirname(_HERE), "include", "c2m_geom.h")


class _Index(dict):
    """Enum of include/c2m_geom.h as attributes: G.M, G.PATCH_TY, ... (unknown names raise -- a typo is not index 0)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"c2m_geom.h defines no geom entry {k!r}") from None


def _parse_geom_header():
    with open(GEOM_HEADER) as f:
        text = f.read()
    version = int(re.search(r"#define\s+C2M_ABI_VERSION\s+(\d+)", text).group(1))
    conv = _Index((m.group(1), int(m.group(2))) for m in re.finditer(r"\bC2M_G_([A-Z0-9_]+)\s*=\s*(\d+)", text))
    wino = _Index((m.group(1), int(m.group(2))) for m in re.finditer(r"\bC2M_WG_([A-Z0-9_]+)\s*=\s*(\d+)", text))
    for tab in (conv, wino):          # every index once
        assert len(set(tab.values())) == len(tab), "c2m_geom.h: two names share an index"
    return version, conv, wino


ABI_VERSION, GEOM, WINO_GEOM = _parse_geom_header()


def declared_symbols():
    """Every function name declared in include/c2m_hip.h."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(?:int|long)\s+(c2m_\w+)\s*\(", text)))


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"c2m_amd: HIP kernel library not found at {LIB_PATH}. Build it with "
                "`python -m c2m_amd.build` (needs hipcc); there is no CPU/PyTorch fallback for these ops.")
        # torch bundles its own HIP runtime (same SONAME as the system one).  It must be in the process BEFORE our
        # library is mapped so that both resolve to ONE runtime; the other order registers our kernels with a second
        # runtime that never sees torch's device context (every launch then fails with hipErrorNoDevice).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if (L.c2m_abi_version(), L.c2m_geom_len(), L.c2m_wino_geom_len()) != (ABI_VERSION, GEOM.LEN, WINO_GEOM.LEN):
            raise RuntimeError(f"c2m_amd: {LIB_PATH} was built from another include/c2m_geom.h (library ABI "
                               f"{L.c2m_abi_version()}, geom {L.c2m_geom_len()} / {L.c2m_wino_geom_len()} entries; header ABI "
                               f"{ABI_VERSION}, {GEOM.LEN} / {WINO_GEOM.LEN}): rebuild with `python -m c2m_amd.build --force`")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"c2m_amd: {what} failed with hipError_t={rc}")
