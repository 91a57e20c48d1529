"""ctypes bindings of the LIC motion compensation: oracle (xo) and reference
harness (xr).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

u16p = C.POINTER(C.c_uint16)
pd = C.c_ssize_t
LIC_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("comp", "u1"),
                      ("neighbors", "u1"), ("mv_x", "<i4"), ("mv_y", "<i4"),
                      ("above_x", "<i2"), ("above_y", "<i2"), ("left_x", "<i2"),
                      ("left_y", "<i2")])
assert LIC_DTYPE.itemsize == 24
HAS_ABOVE, HAS_LEFT = 1, 2


def _planes(planes, borders):
    pp = (u16p * 3)()
    ss = (pd * 3)()
    for c, a in enumerate(planes):
        b = borders[c]
        pp[c] = C.cast(a.ctypes.data + b * a.strides[0] + b * 2, u16p)
        ss[c] = a.strides[0] // 2
    return pp, ss


def xo_mc_lic(xo, bd, job, pic_w, pic_h, ref_planes, borders, rec_planes):
    """ref_planes: padded [Y,U,V] (borders per plane); rec_planes: unpadded.
    Returns the component's prediction plane (zeros outside the block)."""
    f = xo.dll.xo_mc_lic_block
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, u16p, pd, u16p, pd, u16p, pd]
    job = np.ascontiguousarray(job, LIC_DTYPE).reshape(1)
    c = int(job[0]["comp"])
    rp, rs = _planes(ref_planes, borders)
    rec = rec_planes[c]
    pred = np.zeros_like(rec)
    f(bd, job.ctypes.data, pic_w, pic_h, rp[c], rs[c], C.cast(rec.ctypes.data, u16p),
      rec.strides[0] // 2, C.cast(pred.ctypes.data, u16p), pred.strides[0] // 2)
    return pred


def xr_mc_lic(xr, bd, job, above_wh, left_wh, pic_w, pic_h, ref_planes, borders, rec_planes):
    f = xr.dll.xr_mc_lic_block
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p] + [C.c_int] * 6 + \
        [C.POINTER(u16p), C.POINTER(pd), C.POINTER(u16p), C.POINTER(pd), u16p, pd]
    job = np.ascontiguousarray(job, LIC_DTYPE).reshape(1)
    c = int(job[0]["comp"])
    rp, rs = _planes(ref_planes, borders)
    cp, cs = _planes(rec_planes, [0, 0, 0])
    pred = np.zeros_like(rec_planes[c])
    f(bd, job.ctypes.data, above_wh[0], above_wh[1], left_wh[0], left_wh[1], pic_w, pic_h,
      rp, rs, cp, cs, C.cast(pred.ctypes.data, u16p), pred.strides[0] // 2)
    return pred


def random_jobs(rng, pic_w, pic_h, n):
    """(job, above (w,h), left (w,h)) with neighbour CUs of assorted sizes and
    positions (they only have to cover the sample above / left of the CU's
    top-left corner), missing neighbours at picture edges and inside."""
    out = []
    for _ in range(n):
        w, h = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 32, 64]))
        x = int(rng.integers(0, (pic_w - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (pic_h - h) // 8 + 1)) * 8
        if rng.random() < 0.15:
            x = 0
        if rng.random() < 0.15:
            y = 0
        nb, above, left = 0, (8, 8), (8, 8)
        ax = ay = lx = ly = 0
        if y > 0 and rng.random() < 0.9:
            nb |= HAS_ABOVE
            ah = int(rng.choice([8, 16, 32]))
            ah = min(ah, y)
            aw = int(rng.choice([8, 16, 32, 64]))
            ax = max(0, x - int(rng.integers(0, aw // 8)) * 8)
            ay = y - ah
            above = (min(aw, pic_w - ax), ah)
        if x > 0 and rng.random() < 0.9:
            nb |= HAS_LEFT
            lw = min(int(rng.choice([8, 16, 32])), x)
            lh = int(rng.choice([8, 16, 32, 64]))
            ly = max(0, y - int(rng.integers(0, lh // 8)) * 8)
            lx = x - lw
            left = (lw, min(lh, pic_h - ly))
        j = np.zeros(1, LIC_DTYPE)[0]
        j["x"], j["y"], j["w"], j["h"] = x, y, w, h
        j["comp"], j["neighbors"] = int(rng.integers(0, 3)), nb
        j["mv_x"], j["mv_y"] = int(rng.integers(-300, 301)), int(rng.integers(-300, 301))
        if rng.random() < 0.1:      # far outside: exercises every ClipMv
            j["mv_x"], j["mv_y"] = int(rng.integers(-40000, 40001)), int(rng.integers(-40000, 40001))
        j["above_x"], j["above_y"], j["left_x"], j["left_y"] = ax, ay, lx, ly
        out.append((j, above, left))
    return out
