"""Loader for tests/golden/*.npz (written by oracle/capture_golden.py from the live reference)."""
import json
import os

import numpy as np
import torch

from oracle.golden_util import unpack_mask

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.arr = {k: z[k] for k in z.files if k != "meta"}

    def group(self, prefix):
        """{name: tensor} for every array stored as '<prefix>.<name>'."""
        out = {}
        for k, v in self.arr.items():
            if k.startswith(prefix + "."):
                out[k[len(prefix) + 1:]] = torch.from_numpy(np.array(v))
        return out

    def mask(self, name):
        return torch.from_numpy(unpack_mask(self.arr["mask." + name], self.arr["maskshape." + name]))

    def json(self, key):
        return json.loads(bytes(self.arr[key]).decode())


def names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))
