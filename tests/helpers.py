"""Shared test helpers (no GPU needed to import)."""
import ast
import os

import numpy as np
import torch

from egogen_amd.synth import seeded_fill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rebuild_state_dict(g, seeds, prefixes, gains=None, dtype=torch.float32):
    """Rebuild the seeded weights the reference module was filled with (scripts/gen_goldens.py)."""
    keys = [str(k) for k in g["state_dict_keys"]]
    shapes = [ast.literal_eval(str(s)) for s in g["state_dict_shapes"]]
    out = {}
    gains = gains or [1.0] * len(seeds)
    if len(prefixes) == 1 and prefixes[0] == "":
        groups = [("", list(zip(keys, shapes)))]
    else:
        groups = [(p, [(k[len(p):], s) for k, s in zip(keys, shapes) if k.startswith(p)]) for p in prefixes]
    for (p, items), seed, gain in zip(groups, seeds, gains):
        vals = seeded_fill(dict(items), int(seed), gain=gain)
        for k, v in vals.items():
            out[p + k] = torch.from_numpy(v).to(dtype)
    return out


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
