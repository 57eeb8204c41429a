"""TEST INFRASTRUCTURE -- builds the C restatement (oracle/c2m_oracle_index.c) into oracle/_build/.

The Python reference has no compilable sources of its own on the hot path (pure PyTorch), so there is no
"reference compiled here" (`oracle/_ref/`) for this project: `_build/` holds OUR C restatement only, pinned by the
reference-captured fixtures.  Called from __graft_entry__.build(); tests load the .so.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c2m_oracle_index.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle_index.so")
FMA_MODE = 7  # frozen by tests/test_oracle_golden.py::test_c_oracle_fma_mode_is_pinned


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"])
    return LIB


def load():
    return ctypes.CDLL(build())
