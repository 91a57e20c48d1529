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


def order_path(name):
    return os.path.join(GOLDEN, "intra_order_%s.npz" % name)


def load_order(name):
    """tests/golden/intra_order_<clip>.npz (tools/gen_order_golden.py, round 5): EVERY
    DetermineSlowIntraModes call and every TransformAndReconstruct of an intra CU of the
    picture tests/rd_serial.py walks, LM chroma included (mode 67: behind the call's
    reference samples lies the luma rectangle RescaleLuma reads - rows y - 2 .., columns
    x - 3 .. where the picture has them), and where each lies in the order of
    rd_order_<clip>.npz: pos = inter records of that order in front of it, stamp = its
    order among the intra records.  None when the clip has no such fixture."""
    if not os.path.exists(order_path(name)):
        return None
    z = np.load(order_path(name))
    out = {k: z[k] for k in ("samples", "itx_samples")}
    out["contexts"] = z["contexts"]
    out["qps"] = z["qps"]
    for t, dt in (("calls", CALL_DTYPE), ("evals", EVAL_DTYPE), ("itx", ITX_DTYPE)):
        names = [f for f in dt.names if not f.startswith("pad")]
        a = np.zeros(len(z[t + "/" + names[0]]), dt)
        for f in names:
            a[f] = z[t + "/" + f]
        out[t] = a
    out["pos"] = {"calls": z["pos/calls"], "itx": z["pos/itx"]}
    out["stamp"] = {"calls": z["stamp/calls"], "itx": z["stamp/itx"]}
    return out
