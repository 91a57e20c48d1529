"""ctypes bindings of the affine motion estimation: oracle (xo) and reference
harness (xr), plus input generators.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

u16p = C.POINTER(C.c_uint16)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
pd = C.c_ssize_t
BLOCK_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("flags", "u1"),
                        ("reserved", "u1"), ("lambda16", "<u4"), ("mvp", "<i4", (3, 2)),
                        ("bootstrap", "<i4", (3, 2)), ("other_mv", "<i4", (3, 2))])
RESULT_DTYPE = np.dtype([("mv", "<i4", (3, 2)), ("dist", "<u4"), ("iterations", "<u4")])
assert BLOCK_DTYPE.itemsize == 84 and RESULT_DTYPE.itemsize == 32
HAS_BOOTSTRAP, BIPRED = 1, 2


def gradient_search(lib, bd, pred, err):
    """pred: (h, w) uint16, err: (h, w) int16 -> [mvd0.x, mvd0.y, mvd1.x, mvd1.y]."""
    h, w = pred.shape
    pred = np.ascontiguousarray(pred, np.uint16)
    err = np.ascontiguousarray(err, np.int16)
    mvd = np.zeros(4, np.int32)
    if lib.prefix == "xo":
        f = lib.dll.xo_affine_gradient_search
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, u16p, pd, i16p, pd, i32p]
        f(w, h, pred.ctypes.data_as(u16p), w, err.ctypes.data_as(i16p), w,
          mvd.ctypes.data_as(i32p))
    else:
        f = lib.dll.xr_affine_gradient_search
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, C.c_int, u16p, pd, i16p, pd, i32p]
        f(bd, w, h, pred.ctypes.data_as(u16p), w, err.ctypes.data_as(i16p), w,
          mvd.ctypes.data_as(i32p))
    return [int(v) for v in mvd]


def affine_me(lib, bd, block, pic_w, pic_h, orig_pad, ref_pad, border, other_pad=None):
    """orig_pad / ref_pad / other_pad: padded luma planes (other_pad: the other
    list's reference, for BIPRED jobs).  Returns a RESULT_DTYPE scalar."""
    f = getattr(lib.dll, lib.prefix + "_affine_me")
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, u16p, pd, u16p, pd, u16p, pd,
                  C.c_void_p]
    b = np.ascontiguousarray(block, BLOCK_DTYPE).reshape(1)
    out = np.zeros(1, RESULT_DTYPE)
    o = orig_pad[border:, border:]
    r = ref_pad[border:, border:]
    if other_pad is None:
        assert not int(b[0]["flags"]) & BIPRED
        other_pad = ref_pad
    t = other_pad[border:, border:]
    f(bd, b.ctypes.data, pic_w, pic_h, C.cast(o.ctypes.data, u16p), orig_pad.strides[0] // 2,
      C.cast(r.ctypes.data, u16p), ref_pad.strides[0] // 2,
      C.cast(t.ctypes.data, u16p), other_pad.strides[0] // 2, out.ctypes.data)
    return out[0]


def warped_pics(rng, bd, pw, ph, border, zoom=1.0, rot=0.0, shift=(0.0, 0.0), noise=2):
    """ref = a smooth texture + grain; orig = the same texture seen through a
    similarity transform about the picture centre (zoom, rotation in radians,
    shift in samples) -> the gradient iterations have something to find."""
    H, W = ph + 2 * border, pw + 2 * border
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)

    def tex(x, y):
        return (np.sin(x / 6.0) * np.cos(y / 8.0) * 0.25 + np.sin((x + y) / 19.0) * 0.2 +
                np.cos((x - 2 * y) / 31.0) * 0.1 + 0.5)
    mx = (1 << bd) - 1
    ref = np.clip(tex(xx, yy) * mx + rng.integers(-noise, noise + 1, size=(H, W)), 0, mx)
    cx, cy = W / 2.0, H / 2.0
    a, b = zoom * np.cos(rot), zoom * np.sin(rot)
    xs = a * (xx - cx) - b * (yy - cy) + cx + shift[0]
    ys = b * (xx - cx) + a * (yy - cy) + cy + shift[1]
    orig = np.clip(tex(xs, ys) * mx + rng.integers(-noise, noise + 1, size=(H, W)), 0, mx)
    return orig.astype(np.uint16), ref.astype(np.uint16)


def random_blocks(rng, pic_w, pic_h, n, qp_lambda16=(9000, 60000, 400000), bipred=False):
    out = np.zeros(n, BLOCK_DTYPE)
    for i in range(n):
        w, h = int(rng.choice([16, 32, 64])), int(rng.choice([16, 32, 64]))
        x = int(rng.integers(0, (pic_w - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (pic_h - h) // 8 + 1)) * 8
        base = rng.integers(-60, 61, size=2)
        if i % 13 == 0:
            base = rng.integers(-3000, 3001, size=2)      # far: ClipMv at work
        mv0 = base + rng.integers(-8, 9, size=2)
        mv1 = mv0 + (rng.integers(-12, 13, size=2) if i % 4 else 0)
        mv2 = np.array([mv0[0] - (mv1[1] - mv0[1]) * h // w, mv0[1] + (mv1[0] - mv0[0]) * h // w])
        b = out[i]
        b["x"], b["y"], b["w"], b["h"] = x, y, w, h
        b["lambda16"] = int(rng.choice(qp_lambda16))
        b["mvp"] = np.stack([mv0, mv1, mv2])
        if i % 3:
            b["flags"] = HAS_BOOTSTRAP
            if i % 9 == 1:
                b["bootstrap"] = b["mvp"]                 # equal: skipped
            else:
                t = base + rng.integers(-6, 7, size=2)    # translational bootstrap
                b["bootstrap"] = np.stack([t, t, t])
        if bipred and i % 2:
            b["flags"] |= BIPRED
            o0 = -base + rng.integers(-8, 9, size=2)      # roughly mirrored motion
            o1 = o0 + (rng.integers(-12, 13, size=2) if i % 3 else 0)
            o2 = np.array([o0[0] - (o1[1] - o0[1]) * h // w, o0[1] + (o1[0] - o0[0]) * h // w])
            b["other_mv"] = np.stack([o0, o1, o2])
    return out
