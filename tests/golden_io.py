"""Readers for the committed fixtures in tests/golden/ (written by tests/golden/make_goldens.py
from the real reference).  Data only."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unpack_bits(blob, offs):
    """inverse of make_goldens.pack_bits -> list of '01' strings."""
    bits = np.unpackbits(np.asarray(blob, np.uint8))
    s = "".join("01"[b] for b in bits)
    return [s[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]


E2E_FILES = ["e2e_cif_noise.npz", "e2e_qcif_quality.npz", "e2e_360x250.npz", "e2e_400x600.npz", "e2e_1080p.npz"]


def e2e_cases(names=None):
    """Yields (file, qf, mode_name, rgb[n,H,W,C], expected .mpeg bytes)."""
    for fn in names or E2E_FILES:
        z = load(fn)
        for key in z.files:
            if key.startswith("mpeg_"):
                _, mode, q = key.split("_")
                yield fn, int(q[1:]), mode, z["rgb"], z[key].tobytes()
