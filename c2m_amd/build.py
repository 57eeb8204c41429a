"""Builds c2m_amd/lib/libc2m_hip.so (gfx950) from c2m_amd/csrc/*.hip with hipcc.  In-tree, so the .so travels."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libc2m_hip.so")
ARCH = "gfx950"
# index/mask-path files pin the fp32 operation order: no implicit contraction there.
# Every file is built without the SLP vectoriser (NOSLP): with it conv_thin_wgrad_rows_kernel (accumulators packed into
# v_pk_fma_f32 pairs) returned wrong partial sums -- in exactly those pairs -- whenever the bf16 gather kernel ran next to it on
# another stream (tools/dbg_side_stream.py; bit-exact alone; cause not isolated: tools/micro/pk_fma_vs_bf16_mfma.hip does not
# reproduce it with a bare MFMA loop as the neighbour), and the packed form was the SLOWER one on the <= 4-row vector-ALU kernels
# (7x7 head: weight gradient 0.49 -> 0.36 ms, forward 0.34 -> 0.27 ms).  Kernels of the other files can run next to RCCL's on
# the reducer's stream, so they get the same treatment; whole-step time is unchanged (71.18 vs 71.15 ms fp32, 61.9 vs 61.6 bf16).
# The packed fp32 adds of the Winograd kernels are explicit inline asm (they only ever ran next to their own kind in the tests).
NOSLP = ["-fno-slp-vectorize"]
SOURCES = {"conv_igemm.hip": NOSLP, "conv_nc8.hip": NOSLP, "conv_wino.hip": NOSLP, "conv_wino4.hip": NOSLP, "conv_ring.hip": NOSLP, "norm.hip": NOSLP, "losses.hip": NOSLP,
           "optim.hip": ["-ffp-contract=off"] + NOSLP, "data_prep.hip": ["-ffp-contract=off"] + NOSLP,
           "warp.hip": ["-ffp-contract=off"] + NOSLP, "motion_raster.hip": ["-ffp-contract=off"] + NOSLP, "events.hip": [],
           "flownet_ops.hip": ["-ffp-contract=off"] + NOSLP, "gnn.hip": ["-ffp-contract=off"] + NOSLP}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None, defines=()):
    """variant/defines: tuning builds (lib/libc2m_hip_<variant>.so compiled with -D...), selected at run time with
    the environment variable C2M_AMD_LIB; the default build takes neither."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [os.path.join(CSRC, h) for h in ("common.h", "dtype.h", "conv_store.h")] + \
        [os.path.join(os.path.dirname(HERE), "include", "c2m_geom.h")]
    suffix = f"_{variant}" if variant else ""
    lib = LIB.replace(".so", suffix + ".so")
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", suffix + ".o"))
        slp = os.environ.get("C2M_BUILD_SLP", "")         # tuning / bisect builds only (tools/concurrency_stress.py): SLP vectoriser ON
        if slp == "1" or src in slp.split(","):           # for every file ("1") or for the listed ones ("conv_igemm.hip,warp.hip")
            extra = [f for f in extra if f != "-fno-slp-vectorize"]
        if force or variant or _stale(o, [s] + common):
            cmd = [hipcc, "-O3", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden", "-c", s, "-o", o] + extra + \
                  [f"-D{d}" for d in defines]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or variant or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = args[0] if args and not args[0].startswith("-D") else None
    print(build(force="--force" in sys.argv, verbose=True, variant=variant,
                defines=[a[2:] for a in args if a.startswith("-D")]))
