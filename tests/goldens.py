"""Helpers to read tests/golden/*.npz and rebuild the exact weights/inputs the
fixtures were generated with (recipes live in vinet_amd/synth.py)."""
import json
import os

import numpy as np
import torch

from vinet_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


def t(arr):
    return torch.from_numpy(np.asarray(arr))


def state_dict_for(model, weight_seed, z=None, meta=None):
    """procedural weights (+ the calibrated head stored in the fixture)."""
    sd = synth.synth_state_dict(model.state_dict(), weight_seed)
    if z is not None and "head_w" in z.files:
        wk = meta.get("head_w_key")
        bk = meta.get("head_b_key")
        if wk is None:  # decoder-only fixture
            wk, bk = "convtsp4.6.weight", "convtsp4.6.bias"
        sd[wk] = t(z["head_w"])
        sd[bk] = t(z["head_b"])
    return sd
