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


# TransformAndReconstruct calls of intra CUs (oracle/ref_harness.cc, xr_rd::IntraTx)
ITX_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("comp", "u1"),
    ("mode", "u1"), ("neighbors", "u1"), ("above_right", "u1"), ("below_left", "u1"),
    ("tx_skip", "u1"), ("tx_hor", "u1"), ("tx_ver", "u1"), ("scan", "u1"), ("dst4x4", "u1"),
    ("completed", "u1"), ("intra_pic", "u1"), ("qp", "i1"), ("qp_luma", "i1"),
    ("ctx_index", "<i4"), ("qp_index", "<i4"), ("nnz", "<i4"), ("levels_crc", "<u4"),
    ("pred_crc", "<u4"), ("rec_crc", "<u4"), ("sample_off", "<i4"), ("dist", "<u8")], align=True)


def path(name):
    return os.path.join(GOLDEN, "intra_calls_%s.npz" % name)


def load(name):
    z = np.load(path(name))
    out = {"samples": z["samples"]}
    for k in ("itx_samples", "contexts", "qps"):
        if k in z.files:
            out[k] = z[k]
    tables = [("calls", CALL_DTYPE), ("evals", EVAL_DTYPE)]
    if "itx/poc" in z.files:
        tables.append(("itx", ITX_DTYPE))
    for t, dt in tables:
        n = len(z[t + "/" + [f for f in dt.names if not f.startswith("pad")][0]])
        a = np.zeros(n, dt)
        for f in dt.names:
            if not f.startswith("pad"):
                a[f] = z[t + "/" + f]
        out[t] = a
    return out
