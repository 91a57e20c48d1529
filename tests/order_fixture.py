"""tests/golden/rd_order_<clip>.npz (tools/gen_order_golden.py): the global issue
order of the RD search's calls over the me_calls / rd_calls tables, every priced
motion candidate with the bits the encoder's entropy coder gave it, the context
snapshots those bits came from and SearchMotion's final choices.  Record layouts
= the C structs of the capture hooks in oracle/ref_harness.cc (sizes checked by
the generator)."""
import os

import numpy as np

import rd_fixture as rf

GOLDEN = rf.GOLDEN
SEQ_TABLES = ["me", "steps", "merges", "evals", "calls", "cands", "finals"]

CAND_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("kind", "u1"),
    ("flags", "u1"), ("list", "u1"), ("ref_idx", "i1"), ("reused", "u1"), ("mvp_idx", "u1"),
    ("inter_dir", "u1"), ("other_ref_idx", "i1"), ("other_mvp_idx", "u1"),
    ("force_mvd_zero_other", "u1"), ("mv", "<i4", (3, 2)), ("mvp", "<i4", (2, 3, 2)),
    ("start_mvp_idx", "u1"), ("pad", "u1", 3), ("other_mvd", "<i4", (2, 2)), ("dist", "<u4"), ("bits", "<u4"), ("lambda16", "<u4"),
    ("ictx_index", "<i4")], align=True)

FINAL_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("which", "u1"),
    ("flags", "u1"), ("inter_dir", "u1"), ("ref_idx", "i1", 2), ("mvp_idx", "u1", 2),
    ("pad", "u1", 3), ("mv", "<i4", (2, 3, 2)), ("mvd", "<i4", (2, 2, 2))], align=True)

ICTX_DTYPE = np.dtype([
    ("merge_flag", "u1"), ("inter_dir_bi", "u1"), ("inter_dir_l", "u1"), ("affine_flag", "u1"),
    ("ref_idx", "u1", 2), ("mvd", "u1", 2), ("mvp_idx", "u1"), ("fullpel_mv", "u1"),
    ("lic_flag", "u1"), ("flags", "u1"), ("num_refs", "u1", 2), ("frac_bits", "<u2")])

_DTYPES = {"cands": CAND_DTYPE, "finals": FINAL_DTYPE, "ictx": ICTX_DTYPE}


def to_columns(tables):
    out = {}
    for t, a in tables.items():
        for f in a.dtype.names:
            if not f.startswith("pad"):
                out["%s/%s" % (t, f)] = np.ascontiguousarray(a[f])
    return out


def path(name):
    return os.path.join(GOLDEN, "rd_order_%s.npz" % name)


def load(name):
    """-> dict: cands, finals, ictx (record arrays) and seq: {table: uint32 sequence
    number of every record of that table} (tables as in SEQ_TABLES; "me" indexes
    me_calls_<clip>.npz, steps ... calls index rd_calls_<clip>.npz)."""
    z = np.load(path(name))
    out = {}
    for t, dt in _DTYPES.items():
        cols = [k for k in z.files if k.startswith(t + "/")]
        a = np.zeros(len(z[cols[0]]) if cols else 0, dt)
        for k in cols:
            a[k.split("/", 1)[1]] = z[k]
        out[t] = a
    out["seq"] = {t: np.cumsum(z["seq/" + t], dtype=np.int64).astype(np.uint32)
                  for t in SEQ_TABLES}
    return out
