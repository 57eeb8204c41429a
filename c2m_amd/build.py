"""Builds c2m_amd/lib/libc2m_hip.so (gfx950) from c2m_amd/csrc/*.hip with hipcc.  In-tree, so the .so travels."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libc2m_hip.so")
ARCH = "gfx950"
# index/mask-path files pin the fp32 operation order: no implicit contraction there
SOURCES = {"conv_igemm.hip": [], "norm.hip": [], "losses.hip": [],
           "warp.hip": ["-ffp-contract=off"], "motion_raster.hip": ["-ffp-contract=off"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [os.path.join(CSRC, "common.h")]
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + common):
            cmd = [hipcc, "-O3", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden", "-c", s, "-o", o] + extra
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
