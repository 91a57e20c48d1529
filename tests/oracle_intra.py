"""ctypes bindings of the intra prediction functions: the oracle's
(oracle/xvc_oracle_intra.c, prefix xo) and the reference harness's (prefix xr).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
pd = C.c_ssize_t

INTRA_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("comp", "u1"), ("mode", "u1"), ("neighbors", "u1"),
                        ("above_right", "u1"), ("below_left", "u1"), ("reserved", "u1")])
assert INTRA_DTYPE.itemsize == 12
HAS_ABOVE_LEFT, HAS_ABOVE, HAS_LEFT = 1, 2, 4
NUM_MODES = 67


def _p(a):
    return C.cast(a.ctypes.data, u16p), a.strides[0] // 2


def pred_block(lib, prefix, bd, job, rec_plane, pic_w, pic_h):
    """rec_plane: the component plane (no border needed: neighbours the job
    declares available must exist inside it).  Returns the w x h prediction."""
    f = getattr(lib.dll, prefix + "_intra_pred_block")
    f.restype = None
    job = np.ascontiguousarray(job, INTRA_DTYPE).reshape(1)
    pred = np.zeros_like(rec_plane)
    rp, rs = _p(rec_plane)
    pp, ps = _p(pred)
    if prefix == "xo":
        f.argtypes = [C.c_int, C.c_void_p, u16p, pd, u16p, pd]
        f(bd, job.ctypes.data, rp, rs, pp, ps)
    else:
        f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, u16p, pd, u16p, pd]
        f(bd, job.ctypes.data, pic_w, pic_h, rp, rs, pp, ps)
    x, y, w, h = (int(job[0][k]) for k in "xywh")
    out = pred[y:y + h, x:x + w].copy()
    pred[y:y + h, x:x + w] = 0
    assert not pred.any()        # nothing written outside the block
    return out


def satd_modes(lib, prefix, bd, job, orig, rec):
    f = getattr(lib.dll, prefix + "_intra_satd_modes")
    f.restype = None
    job = np.ascontiguousarray(job, INTRA_DTYPE).reshape(1)
    dist = np.zeros(NUM_MODES, np.uint32)
    op, os_ = _p(orig)
    rp, rs = _p(rec)
    h, w = orig.shape
    dp = C.cast(dist.ctypes.data, u32p)
    if prefix == "xo":
        f.argtypes = [C.c_int, C.c_void_p, u16p, pd, u16p, pd, u32p]
        f(bd, job.ctypes.data, op, os_, rp, rs, dp)
    else:
        f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, u16p, pd, u16p, pd, u32p]
        f(bd, job.ctypes.data, w, h, op, os_, rp, rs, dp)
    return dist


def random_jobs(rng, pic_w, pic_h, comp, n, sizes=(4, 8, 16, 32, 64)):
    """Blocks anywhere in a pic_w x pic_h component plane with every kind of
    neighbour availability the CU map can produce (picture edges, not yet
    coded above-right / below-left areas, partial counts in steps of 4 luma /
    2 chroma samples)."""
    jobs = np.zeros(n, INTRA_DTYPE)
    step = 4 if comp == 0 else 2
    for j in jobs:
        while True:
            w, h = int(rng.choice(sizes)), int(rng.choice(sizes))
            if w <= pic_w and h <= pic_h and max(w, h) <= 4 * min(w, h) * 4:
                break
        x = int(rng.integers(0, (pic_w - w) // step + 1)) * step
        y = int(rng.integers(0, (pic_h - h) // step + 1)) * step
        if rng.random() < 0.2:
            x = 0
        if rng.random() < 0.2:
            y = 0
        nb = 0
        if x > 0:
            nb |= HAS_LEFT
        if y > 0:
            nb |= HAS_ABOVE
        if x > 0 and y > 0:
            nb |= HAS_ABOVE_LEFT
        ar = bl = 0
        if y > 0:   # GetCuSizeAboveRight: 0..h, clipped by the picture
            room = max(0, min(h, pic_w - (x + w)))
            ar = int(rng.choice([0, room, int(rng.integers(0, room // step + 1)) * step]))
        if x > 0:
            room = max(0, min(w, pic_h - (y + h)))
            bl = int(rng.choice([0, room, int(rng.integers(0, room // step + 1)) * step]))
        j["x"], j["y"], j["w"], j["h"], j["comp"] = x, y, w, h, comp
        j["neighbors"], j["above_right"], j["below_left"] = nb, ar, bl
        j["mode"] = int(rng.integers(0, NUM_MODES))
    return jobs


def lm_chroma(lib, prefix, bd, comp, x, y, w, h, planes):
    """LM chroma prediction of the (x, y, w, h) block (chroma samples) of
    component comp from the reconstruction planes [Y, U, V]."""
    out = np.zeros((h, w), np.uint16)
    if prefix == "xo":
        f = lib.dll.xo_intra_lm_chroma
        f.restype = None
        f.argtypes = [C.c_int] * 5 + [u16p, pd, u16p, pd, u16p, pd]
        lp, ls = _p(planes[0])
        cp, cs = _p(planes[comp])
        f(bd, x, y, w, h, lp, ls, cp, cs, C.cast(out.ctypes.data, u16p), w)
    else:
        f = lib.dll.xr_intra_lm_chroma
        f.restype = None
        f.argtypes = [C.c_int] * 8 + [C.POINTER(u16p), C.POINTER(pd), u16p, pd]
        pp = (u16p * 3)()
        ss = (pd * 3)()
        for c in range(3):
            pp[c], ss[c] = _p(planes[c])
        ph, pw = planes[0].shape
        f(bd, comp, x, y, w, h, pw, ph, pp, ss, C.cast(out.ctypes.data, u16p), w)
    return out
