"""tests/golden/rd_calls_<clip>.npz: the bi-prediction / affine / merge /
TransformAndReconstruct calls the reference encoder's RD search made while it
coded the clips of the stream fixtures (tools/gen_rd_golden.py).  Record
layouts (C struct layout of the capture hooks in oracle/ref_harness.cc; sizes
checked by the generator), a loader, and the host-side preparation of the
device batches that replay them."""
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KIND_BI, KIND_AFFINE_UNI, KIND_AFFINE_BI = 1, 2, 3
FLAG_FULLPEL, FLAG_LIC, FLAG_HAS_BOOT, FLAG_AFFINE, FLAG_MERGE = 1, 2, 4, 8, 16
CTX_BYTES = 152

STEP_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("kind", "u1"),
    ("flags", "u1"), ("list", "u1"), ("ref_idx", "i1"), ("other_ref_idx", "i1"),
    ("start_mvp_idx", "u1"), ("final_mvp_idx", "u1"), ("pad", "u1", 3),
    ("ref_poc", "<i4"), ("other_ref_poc", "<i4"), ("lambda16", "<u4"),
    ("mvp", "<i4", (2, 3, 2)), ("boot", "<i4", (3, 2)), ("other_mv", "<i4", (3, 2)),
    ("mv", "<i4", (3, 2)), ("dist", "<u4"), ("nb_index", "<i4")], align=True)

MERGE_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("pad", "u1", 2),
    ("lambda_sqrt", "<f8"), ("inter_dir", "u1", 5), ("use_lic", "u1", 5),
    ("ref_idx", "i1", (5, 2)), ("ref_poc", "<i4", (5, 2)), ("mv", "<i4", (5, 2, 2)),
    ("order", "<i4", 5), ("cost", "<f8", 5), ("num", "<i4"), ("nb_index", "<i4")], align=True)

EVAL_DTYPE = np.dtype([
    ("poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("inter_dir", "u1"),
    ("flags", "u1"), ("ref_idx", "i1", 2), ("qp", "i1", 3), ("pad", "u1"),
    ("ref_poc", "<i4", 2), ("mv", "<i4", (2, 3, 2)), ("ctx_index", "<i4"), ("qp_index", "<i4"),
    ("nb_index", "<i4"), ("pad2", "<i4"), ("dist_zero", "<u8", 3)], align=True)

QP_DTYPE = np.dtype([
    ("qp_raw", "i1", 3), ("pad", "u1", 5), ("lambda", "<i8", 3), ("rd_factor", "<i8", 3),
    ("dist_weight", "<f8", 3)], align=True)

CALL_DTYPE = np.dtype([
    ("eval", "<i4"), ("comp", "u1"), ("tx_skip", "u1"), ("tx_hor", "u1"), ("tx_ver", "u1"),
    ("scan", "u1"), ("completed", "u1"), ("tx_select_idx", "i1"), ("pad", "u1"),
    ("nnz", "<i4"), ("levels_crc", "<u4"), ("rec_crc", "<u4"), ("pad2", "<u4"),
    ("dist", "<u8")], align=True)

NB_DTYPE = np.dtype([
    ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("has_above", "u1"), ("has_left", "u1"),
    ("above_x", "<i2"), ("above_y", "<i2"), ("left_x", "<i2"), ("left_y", "<i2"),
    ("sample_off", "<u4"), ("sample_count", "<u4")], align=True)

_DTYPES = {"neighbours": NB_DTYPE, "steps": STEP_DTYPE, "merges": MERGE_DTYPE, "evals": EVAL_DTYPE, "qps": QP_DTYPE,
           "calls": CALL_DTYPE}


def path(name):
    return os.path.join(GOLDEN, "rd_calls_%s.npz" % name)


# on disk the tables are stored field by field (deflate works far better on
# columns than on interleaved records), evals' indices in `calls` as deltas,
# the reconstruction CRC truncated to 16 bits
def to_columns(tables):
    out = {}
    for t, a in tables.items():
        if a.dtype.names is None:
            out[t] = a
            continue
        for f in a.dtype.names:
            if f.startswith("pad"):
                continue
            col = np.ascontiguousarray(a[f])
            if (t, f) == ("calls", "eval"):
                col = np.diff(col, prepend=np.int32(0)).astype(np.int32)
            if (t, f) == ("calls", "rec_crc"):
                col = (col & 0xffff).astype(np.uint16)
            out["%s/%s" % (t, f)] = col
    return out


def load(name):
    """-> dict of record arrays (steps, merges, evals, qps, calls, neighbours) +
    contexts [n, 152] uint8 + nb_samples uint16.  calls["rec_crc"] holds the low
    16 bits of the reconstruction block's CRC-32."""
    z = np.load(path(name))
    out = {}
    for t, dt in _DTYPES.items():
        cols = [k for k in z.files if k.startswith(t + "/")]
        n = len(z[cols[0]]) if cols else 0
        a = np.zeros(n, dt)
        for k in cols:
            f = k.split("/", 1)[1]
            col = z[k]
            if (t, f) == ("calls", "eval"):
                col = np.cumsum(col, dtype=np.int64).astype(np.int32)
            a[f] = col
        out[t] = a
    out["contexts"] = z["contexts"].reshape(-1, CTX_BYTES)
    out["nb_samples"] = z["nb_samples"]
    return out


def crc32_rows(block):
    """CRC-32 (zlib polynomial) of a 2-D int16 / uint16 block, row-major - what
    the capture hook computed for the levels and the reconstruction."""
    return zlib.crc32(np.ascontiguousarray(block).tobytes()) & 0xffffffff


def layers(x, y, w, h):
    """Greedy partition of blocks into batches whose members do not overlap
    (the device entry points write a job's prediction / reconstruction at the
    CU's own position in a picture): -> list of index arrays.  Blocks are on
    the 4-sample grid; a batch keeps a cell-occupancy bitmap."""
    n = len(x)
    x4, y4 = np.asarray(x) // 4, np.asarray(y) // 4
    w4, h4 = (np.asarray(w) + 3) // 4, (np.asarray(h) + 3) // 4
    gw, gh = int((x4 + w4).max()) + 1, int((y4 + h4).max()) + 1
    maps, members = [], []
    for i in range(n):
        sl = (slice(y4[i], y4[i] + h4[i]), slice(x4[i], x4[i] + w4[i]))
        for k, m in enumerate(maps):
            if not m[sl].any():
                m[sl] = True
                members[k].append(i)
                break
        else:
            m = np.zeros((gh, gw), bool)
            m[sl] = True
            maps.append(m)
            members.append([i])
    return [np.array(m, np.int64) for m in members]
