"""Helpers shared by the test modules."""
import os

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def coo_of(g, which="sorted"):
    """COO of a fixture as a fresh writable structured array (struct mtx layout)."""
    from spmv_amd.matrix import make_coo
    return make_coo(g[f"coo_{which}_i"], g[f"coo_{which}_j"], g[f"coo_{which}_a"])


def bits_equal(a, b) -> bool:
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()
