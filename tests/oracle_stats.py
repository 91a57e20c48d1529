"""ctypes bindings of the whole-picture passes around the hot path: the
oracle's (oracle/xvc_oracle_stats.c, prefix xo) and the reference harness's
(oracle/ref_harness.cc, prefix xr).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

import oracle_lib as ol

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u64p = C.POINTER(C.c_uint64)
pd = C.c_ssize_t


def _planes(planes):
    pp = (u16p * 3)()
    ss = (pd * 3)()
    for c, a in enumerate(planes):
        assert a.dtype == np.uint16 and a.strides[1] == 2
        pp[c] = C.cast(a.ctypes.data, u16p)
        ss[c] = a.strides[0] // 2
    return pp, ss


def chroma_dims(w, h):
    return [(w, h), (w >> 1, h >> 1), (w >> 1, h >> 1)]


def pack_input(planes, in_bd):
    """[Y,U,V] arrays -> the packed planar byte string an application hands
    over (1 byte per sample at 8 bits, else 2 little endian)."""
    dt = np.uint8 if in_bd == 8 else np.dtype("<u2")
    return b"".join(np.ascontiguousarray(p, dt).tobytes() for p in planes)


# ---- oracle ----
def xo_import_picture(xo, in_bd, out_bd, in_w, in_h, out_w, out_h, data):
    f = xo.dll.xo_import_plane
    f.restype = None
    f.argtypes = [C.c_int] * 6 + [C.c_char_p, pd, u16p, pd]
    bps = 1 if in_bd == 8 else 2
    out, off = [], 0
    for (iw, ih), (ow, oh) in zip(chroma_dims(in_w, in_h), chroma_dims(out_w, out_h)):
        dst = np.zeros((oh, ow), np.uint16)
        f(in_bd, out_bd, iw, ih, ow, oh, data[off:off + iw * ih * bps], iw * bps,
          ol.ptr(dst, u16p), ow)
        off += iw * ih * bps
        out.append(dst)
    return out


def xo_export_picture(xo, bd, out_bd, dither, planes, disp_w, disp_h):
    f = xo.dll.xo_export_plane
    f.restype = None
    f.argtypes = [C.c_int] * 5 + [u16p, pd, u8p]
    bps = 1 if out_bd <= 8 else 2
    out = b""
    for p, (w, h) in zip(planes, chroma_dims(disp_w, disp_h)):
        buf = np.zeros(w * h * bps, np.uint8)
        f(bd, out_bd, int(dither), w, h, C.cast(p.ctypes.data, u16p), p.strides[0] // 2,
          ol.ptr(buf, u8p))
        out += buf.tobytes()
    return out


def _crc(dll, name, bd, mode, w, h, planes):
    f = getattr(dll, name)
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 4 + [C.POINTER(u16p), C.POINTER(pd), u8p]
    pp, ss = _planes(planes)
    hash_ = np.zeros(8, np.uint8)
    n = f(bd, mode, w, h, pp, ss, ol.ptr(hash_, u8p))
    return hash_[:n].tobytes()


def xo_picture_crc(xo, bd, mode, w, h, planes):
    return _crc(xo.dll, "xo_picture_crc", bd, mode, w, h, planes)


def xr_picture_crc(xr, bd, mode, w, h, planes):
    return _crc(xr.dll, "xr_picture_crc", bd, mode, w, h, planes)


def xo_variance_map(xo, w, h, luma):
    """luma: array whose [0:h,0:w] is the picture and which extends at least to
    the next multiple of 16 in both directions."""
    f = xo.dll.xo_variance_map
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, u16p, pd, u64p]
    out = np.zeros(((h + 15) // 16, (w + 15) // 16), np.uint64)
    f(w, h, C.cast(luma.ctypes.data, u16p), luma.strides[0] // 2, ol.ptr(out, u64p))
    return out


def xo_ctu_variance(xo, w, h, x, y, ctu, var_map):
    f = xo.dll.xo_ctu_variance
    f.restype = C.c_uint64
    f.argtypes = [C.c_int] * 5 + [u64p]
    vm = np.ascontiguousarray(var_map, np.uint64)
    return int(f(w, h, x, y, ctu, ol.ptr(vm, u64p)))


def xo_aqp_delta_qp(xo, var, bd, strength):
    f = xo.dll.xo_aqp_delta_qp
    f.restype = C.c_int
    f.argtypes = [C.c_uint64, C.c_int, C.c_int]
    return f(var, bd, strength)


def xo_histogram_distance(xo, bd, a, b):
    f = xo.dll.xo_histogram_distance
    f.restype = C.c_int64
    f.argtypes = [C.c_int] * 3 + [u16p, pd, u16p, pd]
    h, w = a.shape
    return int(f(bd, w, h, C.cast(a.ctypes.data, u16p), a.strides[0] // 2,
                 C.cast(b.ctypes.data, u16p), b.strides[0] // 2))


def xo_allow_lic(xo, dist, w, h):
    f = xo.dll.xo_allow_lic
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.c_int, C.c_int]
    return f(dist, w, h)


# ---- reference harness ----
def xr_import_picture(xr, in_bd, out_bd, in_w, in_h, out_w, out_h, data):
    f = xr.dll.xr_import_picture
    f.restype = None
    f.argtypes = [C.c_int] * 6 + [C.c_char_p, u16p]
    flat = np.zeros(out_w * out_h * 3 // 2, np.uint16)
    f(in_bd, out_bd, in_w, in_h, out_w, out_h, data, ol.ptr(flat, u16p))
    out, off = [], 0
    for w, h in chroma_dims(out_w, out_h):
        out.append(flat[off:off + w * h].reshape(h, w))
        off += w * h
    return out


def xr_export_picture(xr, bd, out_bd, dither, planes, disp_w, disp_h):
    f = xr.dll.xr_export_picture
    f.restype = C.c_size_t
    f.argtypes = [C.c_int] * 7 + [C.POINTER(u16p), C.POINTER(pd), u8p]
    h, w = planes[0].shape
    pp, ss = _planes(planes)
    buf = np.zeros(disp_w * disp_h * 3, np.uint8)
    n = f(bd, out_bd, int(dither), w, h, disp_w, disp_h, pp, ss, ol.ptr(buf, u8p))
    return buf[:n].tobytes()


def xr_aqp_delta_qp(xr, bd, luma, x, y, ctu, strength):
    f = xr.dll.xr_aqp_delta_qp
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 3 + [u16p, pd] + [C.c_int] * 4
    h, w = luma.shape
    return f(bd, w, h, C.cast(luma.ctypes.data, u16p), luma.strides[0] // 2, x, y, ctu,
             strength)


def xr_allow_lic(xr, bd, a, b):
    f = xr.dll.xr_allow_lic
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 3 + [u16p, pd, u16p, pd]
    h, w = a.shape
    return f(bd, w, h, C.cast(a.ctypes.data, u16p), a.strides[0] // 2,
             C.cast(b.ctypes.data, u16p), b.strides[0] // 2)
