"""tests/golden/intra_calls_*.npz (tools/gen_intra_golden.py): a sample of the
reference encoder's IntraSearch::DetermineSlowIntraModes calls - CU, neighbour
state, the reconstruction's row above / column left at that moment, and the SATD
of every mode it evaluated."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CALL_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("neighbors", "u1"),
    ("above_right", "u1"), ("below_left", "u1"), ("pad", "u1", 3), ("sample_off", "<i4"),
    ("first_eval", "<i4"), ("n_eval", "<i4")], align=True)
EVAL_DTYPE = np.dtype([("call", "<i4"), ("dist", "<u4"), ("mode", "u1"), ("pad", "u1", 3)],
                      align=True)


def path(name):
    return os.path.join(GOLDEN, "intra_calls_%s.npz" % name)


def load(name):
    z = np.load(path(name))
    out = {"samples": z["samples"]}
    for t, dt in (("calls", CALL_DTYPE), ("evals", EVAL_DTYPE)):
        n = len(z[t + "/" + [f for f in dt.names if not f.startswith("pad")][0]])
        a = np.zeros(n, dt)
        for f in dt.names:
            if not f.startswith("pad"):
                a[f] = z[t + "/" + f]
        out[t] = a
    return out
