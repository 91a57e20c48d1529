"""ctypes bindings for the CPU oracle (oracle/_build/libxvcoracle.so) and, when
present, the reference harness (oracle/_ref/libxvcref.so).

TEST INFRASTRUCTURE ONLY -- never imported by the product package xvc_amd.
Both libraries expose the same signatures with prefix ``xo_`` / ``xr_``; the
:class:`Lib` wrapper takes the prefix so a test can run the same call against
either one.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libxvcoracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libxvcref.so")

u16p = C.POINTER(C.c_uint16)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
pd = C.c_ssize_t


class CuInfo(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint8),
                ("h", C.c_uint8), ("intra", C.c_uint8), ("cbf_luma", C.c_uint8),
                ("qp_y", C.c_int8), ("qp_c", C.c_int8), ("ref_idx0", C.c_int8),
                ("reserved", C.c_int8), ("ref_poc", C.c_int32 * 2),
                ("mv", C.c_int32 * 2 * 4 * 2)]


class MeBlock(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_uint8),
                ("h", C.c_uint8), ("depth_nonzero", C.c_uint8),
                ("fullpel_mv", C.c_uint8), ("mvp_x", C.c_int32),
                ("mvp_y", C.c_int32), ("prev_x", C.c_int32),
                ("prev_y", C.c_int32), ("lambda16", C.c_uint32),
                ("search_range", C.c_int32)]


class MeResult(C.Structure):
    _fields_ = [("fullpel_x", C.c_int32), ("fullpel_y", C.c_int32),
                ("mv_x", C.c_int32), ("mv_y", C.c_int32),
                ("fullpel_cost", C.c_uint32), ("subpel_dist", C.c_uint32)]


class BiBlock(C.Structure):
    _fields_ = [("blk", MeBlock), ("other_mv_x", C.c_int32),
                ("other_mv_y", C.c_int32), ("boot_mv_x", C.c_int32),
                ("boot_mv_y", C.c_int32)]


class McBiBlock(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_uint8),
                ("h", C.c_uint8), ("comp", C.c_uint8), ("reserved", C.c_uint8),
                ("mv0_x", C.c_int32), ("mv0_y", C.c_int32),
                ("mv1_x", C.c_int32), ("mv1_y", C.c_int32)]


class TxBlock(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_uint8),
                ("h", C.c_uint8), ("comp", C.c_uint8), ("tx_hor", C.c_uint8),
                ("tx_ver", C.c_uint8), ("dst4x4", C.c_uint8), ("qp", C.c_int8),
                ("intra_pic", C.c_uint8)]


class McBlock(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_uint8),
                ("h", C.c_uint8), ("comp", C.c_uint8), ("reserved", C.c_uint8),
                ("mv_x", C.c_int32), ("mv_y", C.c_int32)]


CU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"),
                     ("intra", "u1"), ("cbf_luma", "u1"), ("qp_y", "i1"),
                     ("qp_c", "i1"), ("ref_idx0", "i1"), ("reserved", "i1"),
                     ("ref_poc", "<i4", (2,)), ("mv", "<i4", (2, 4, 2))])
ME_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("depth_nonzero", "u1"), ("fullpel_mv", "u1"),
                     ("mvp_x", "<i4"), ("mvp_y", "<i4"), ("prev_x", "<i4"),
                     ("prev_y", "<i4"), ("lambda16", "<u4"),
                     ("search_range", "<i4")])
MERES_DTYPE = np.dtype([("fullpel_x", "<i4"), ("fullpel_y", "<i4"),
                        ("mv_x", "<i4"), ("mv_y", "<i4"),
                        ("fullpel_cost", "<u4"), ("subpel_dist", "<u4")])
BI_DTYPE = np.dtype([("blk", ME_DTYPE), ("other_mv_x", "<i4"),
                     ("other_mv_y", "<i4"), ("boot_mv_x", "<i4"),
                     ("boot_mv_y", "<i4")])
MCBI_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                       ("comp", "u1"), ("reserved", "u1"), ("mv0_x", "<i4"),
                       ("mv0_y", "<i4"), ("mv1_x", "<i4"), ("mv1_y", "<i4")])
MCAFF_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("comp", "u1"), ("reserved", "u1"), ("mv", "<i4", (3, 2))])
MCM_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                      ("metric", "u1"), ("qp", "i1"), ("mv_x", "<i4"), ("mv_y", "<i4")])
TX_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("comp", "u1"), ("tx_hor", "u1"), ("tx_ver", "u1"),
                     ("dst4x4", "u1"), ("qp", "i1"), ("intra_pic", "u1")])
MC_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("comp", "u1"), ("reserved", "u1"), ("mv_x", "<i4"),
                     ("mv_y", "<i4")])
assert CU_DTYPE.itemsize == C.sizeof(CuInfo) == 84
assert ME_DTYPE.itemsize == C.sizeof(MeBlock) == 32
assert MERES_DTYPE.itemsize == C.sizeof(MeResult) == 24
assert TX_DTYPE.itemsize == C.sizeof(TxBlock) == 12
assert MC_DTYPE.itemsize == C.sizeof(McBlock) == 16

# Chroma qp mapping for 4:2:0 with chroma_qp_offset_table == 1
# (Qp::kChromaScale_, quantize.cc:34-38): identity below 30, then the
# HEVC-style compression; host-side helper for building xvcgpu_cu_info.
CHROMA_SCALE = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36,
                                  36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45,
                                  46, 47, 48, 49, 50, 51]


def chroma_qp(qp):
    return CHROMA_SCALE[max(0, min(57, qp))]


def build_oracle():
    """(Re)build the oracle .so with gcc if it is missing or stale."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
            if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "xvcgpu_types.h"))
    if os.path.exists(ORACLE_SO):
        so_m = os.path.getmtime(ORACLE_SO)
        if all(os.path.getmtime(s) <= so_m for s in srcs):
            return ORACLE_SO
    # one builder at a time (pytest-xdist workers all arrive here together: a
    # second make would rewrite the .so while the first worker is loading it)
    import fcntl
    os.makedirs(os.path.dirname(ORACLE_SO), exist_ok=True)
    with open(ORACLE_SO + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        stale = not os.path.exists(ORACLE_SO) or \
            any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
        if stale:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def ptr(a, typ):
    return a.ctypes.data_as(typ)


class Lib:
    """Uniform wrapper: Lib('xo') = oracle, Lib('xr') = reference harness."""

    def __init__(self, prefix):
        self.prefix = prefix
        if prefix == "xo":
            self.dll = C.CDLL(build_oracle())
        else:
            self.dll = C.CDLL(REF_SO)
        d, p = self.dll, prefix

        def sig(name, res, args):
            f = getattr(d, p + "_" + name)
            f.restype = res
            f.argtypes = args
            setattr(self, "_" + name, f)

        margs = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                 C.c_int]
        sig("metric_ss", C.c_uint64, margs + [u16p, pd, u16p, pd])
        sig("metric_rs", C.c_uint64, margs + [i16p, pd, u16p, pd])
        sig("ssd_rr", C.c_uint64,
            [C.c_int, C.c_double, C.c_int, C.c_int, i16p, pd, i16p, pd])
        sig("picture_ssd", C.c_uint64,
            [C.c_int, C.c_int, C.c_int, u16p, pd, u16p, pd, u64p, u64p])
        if prefix == "xo":  # sharding helper; the reference only walks whole pictures
            sig("picture_ssd_rows", C.c_uint64,
                [C.c_int] * 5 + [u16p, pd, u16p, pd, u64p, u64p])
        fa = [C.c_int] * 6
        sig("mc_uni", None, fa + [u16p, pd, u16p, pd])
        sig("mc_uni_bipred", None, fa + [u16p, pd, i16p, pd])
        sig("add_avg", None,
            [C.c_int, C.c_int, C.c_int, i16p, pd, i16p, pd, u16p, pd])
        sig("clip_mv", None, [C.c_int] * 4 + [i32p, i32p])
        sig("mc_block", None, [C.c_int] * 10 + [u16p, pd, u16p, pd])
        sig("fwd_transform", None, [C.c_int] * 6 + [i16p, pd, i16p, pd])
        sig("inv_transform", None, [C.c_int] * 7 + [i16p, pd, i16p, pd])
        if prefix == "xr":  # restricted mode (disable_ext2_transform_high_precision)
            sig("fwd_transform_restricted", None, [C.c_int] * 6 + [i16p, pd, i16p, pd])
            sig("inv_transform_restricted", None, [C.c_int] * 7 + [i16p, pd, i16p, pd])
        sig("fwd_transform_skip", None, [C.c_int] * 3 + [i16p, pd, i16p, pd])
        sig("inv_transform_skip", None, [C.c_int] * 3 + [i16p, pd, i16p, pd])
        sig("dequant", None, [C.c_int] * 4 + [i16p, pd, i16p, pd])
        sig("quant_fast", C.c_int, [C.c_int] * 5 + [i16p, pd, i16p, pd])
        sig("transform_matrix", i16p, [C.c_int, C.c_int])
        sig("mvd_bits_fullpel", C.c_uint32, [C.c_int] * 5)
        sig("mvd_bits", C.c_uint32, [C.c_int] * 5)
        sig("min_max_mv", None, [C.c_int] * 7 + [i32p, i32p])
        sig("tz_search", None,
            [C.c_int, C.POINTER(MeBlock), C.c_int, C.c_int, u16p, pd, u16p, pd,
             i32p, u32p])
        sig("subpel_search", None,
            [C.c_int, C.POINTER(MeBlock), C.c_int, C.c_int, u16p, pd, u16p, pd,
             i32p, i32p, u32p])
        sig("bipred_search", None,
            [C.c_int, C.POINTER(BiBlock), C.c_int, C.c_int, u16p, pd, u16p, pd,
             u16p, pd, C.POINTER(MeResult)])
        sig("quant_fast2", C.c_int, [C.c_int] * 7 + [i16p, pd, i16p, pd])
        sig("mc_affine_block", None,
            [C.c_int] * 6 + [i32p, C.c_int, C.c_int, u16p, pd, u16p, pd])
        sig("mc_metric", C.c_uint64,
            [C.c_int] * 12 + [u16p, pd, u16p, pd])
        sig("mc_bipred_block", None,
            [C.c_int] * 12 + [u16p, pd, u16p, pd, u16p, pd])
        if p == "xo":
            sig("full_search", None,
                [C.c_int] * 8 + [C.c_uint32, i32p, i32p, i16p, pd, u16p, pd,
                                 i32p])
            sig("deblock_picture", None,
                [C.c_int] * 7 + [C.POINTER(CuInfo), i32p, C.c_int,
                                 C.POINTER(u16p), C.POINTER(pd)])
            sig("pad_border", None, [C.c_int] * 4 + [u16p, pd])
            sig("residual_pipeline", C.c_int,
                [C.c_int, C.POINTER(TxBlock), u16p, pd, u16p, pd, u16p, pd,
                 i16p])
        else:
            sig("full_search", None,
                [C.c_int] * 8 + [C.c_uint32, i32p, i32p, i16p, pd, u16p, pd,
                                 C.c_int, C.c_int, i32p])
            sig("deblock_picture", None,
                [C.c_int] * 7 + [C.POINTER(CuInfo), C.c_int, C.POINTER(u16p),
                                 C.POINTER(pd), i32p, C.c_int, i32p, C.c_int])
            sig("pad_border", None,
                [C.c_int, C.c_int, C.POINTER(u16p), C.POINTER(pd)])
            sig("set_simd", None, [C.c_int])

    # ---- numpy-friendly wrappers (2-D arrays, element strides) ----
    @staticmethod
    def _s(a):
        return a.strides[0] // a.itemsize

    def metric_ss(self, metric, bd, a, b, qp=32, strength=16, weight=1.0):
        h, w = a.shape
        return self._metric_ss(metric, bd, qp, strength, weight, w, h,
                               ptr(a, u16p), self._s(a), ptr(b, u16p),
                               self._s(b))

    def metric_rs(self, metric, bd, a, b, qp=32, strength=16, weight=1.0):
        h, w = a.shape
        return self._metric_rs(metric, bd, qp, strength, weight, w, h,
                               ptr(a, i16p), self._s(a), ptr(b, u16p),
                               self._s(b))

    def ssd_rr(self, bd, a, b):
        h, w = a.shape
        return self._ssd_rr(bd, 1.0, w, h, ptr(a, i16p), self._s(a),
                            ptr(b, i16p), self._s(b))

    def picture_ssd(self, bd, a, b, y_begin=0, y_end=1 << 30):
        h, w = a.shape
        d = C.c_uint64(0)
        n = C.c_uint64(0)
        if y_begin <= 0 and y_end >= h:
            r = self._picture_ssd(bd, w, h, ptr(a, u16p), self._s(a), ptr(b, u16p),
                                  self._s(b), C.byref(d), C.byref(n))
        else:
            r = self._picture_ssd_rows(bd, w, h, y_begin, y_end, ptr(a, u16p), self._s(a),
                                       ptr(b, u16p), self._s(b), C.byref(d), C.byref(n))
        return r, n.value

    def mc_uni(self, bd, is_chroma, w, h, fx, fy, plane, px, py, bipred=False):
        """plane: padded 2-D array; (px,py): full-pel position inside it."""
        out = np.zeros((h, w), np.int16 if bipred else np.uint16)
        ref = plane[py:, px:]
        f = self._mc_uni_bipred if bipred else self._mc_uni
        f(bd, is_chroma, w, h, fx, fy, ptr(ref, u16p), self._s(plane),
          ptr(out, i16p if bipred else u16p), w)
        return out

    def add_avg(self, bd, a, b):
        h, w = a.shape
        out = np.zeros((h, w), np.uint16)
        self._add_avg(bd, w, h, ptr(a, i16p), self._s(a), ptr(b, i16p),
                      self._s(b), ptr(out, u16p), w)
        return out

    def clip_mv(self, x, y, pw, ph, mx, my):
        a, b = C.c_int32(mx), C.c_int32(my)
        self._clip_mv(x, y, pw, ph, C.byref(a), C.byref(b))
        return a.value, b.value

    def mc_block(self, bd, comp, x, y, w, h, mx, my, pw, ph, padded, border):
        """padded: plane incl. `border` samples on each side (component res)."""
        cs = 1 if comp else 0
        out = np.zeros((h >> cs, w >> cs), np.uint16)
        origin = padded[border:, border:]
        self._mc_block(bd, comp, x, y, w, h, mx, my, pw, ph, ptr(origin, u16p),
                       self._s(padded), ptr(out, u16p), w >> cs)
        return out

    def fwd_transform(self, bd, resi, tx_hor=0, tx_ver=0, dst4x4=0):
        h, w = resi.shape
        out = np.zeros((h, w), np.int16)
        self._fwd_transform(bd, w, h, tx_hor, tx_ver, dst4x4, ptr(resi, i16p),
                            self._s(resi), ptr(out, i16p), w)
        return out

    def inv_transform(self, bd, coeff, tx_hor=0, tx_ver=0, dst4x4=0, dc_only=0):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        self._inv_transform(bd, w, h, tx_hor, tx_ver, dst4x4, dc_only,
                            ptr(coeff, i16p), self._s(coeff), ptr(out, i16p), w)
        return out

    def fwd_transform_restricted(self, bd, resi, tx_hor=0, tx_ver=0, dst4x4=0):
        h, w = resi.shape
        out = np.zeros((h, w), np.int16)
        self._fwd_transform_restricted(bd, w, h, tx_hor, tx_ver, dst4x4, ptr(resi, i16p),
                                       self._s(resi), ptr(out, i16p), w)
        return out

    def inv_transform_restricted(self, bd, coeff, tx_hor=0, tx_ver=0, dst4x4=0, dc_only=0):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        self._inv_transform_restricted(bd, w, h, tx_hor, tx_ver, dst4x4, dc_only,
                                       ptr(coeff, i16p), self._s(coeff), ptr(out, i16p), w)
        return out

    def fwd_transform_skip(self, bd, resi):
        h, w = resi.shape
        out = np.zeros((h, w), np.int16)
        self._fwd_transform_skip(bd, w, h, ptr(resi, i16p), self._s(resi),
                                 ptr(out, i16p), w)
        return out

    def inv_transform_skip(self, bd, coeff):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        self._inv_transform_skip(bd, w, h, ptr(coeff, i16p), self._s(coeff),
                                 ptr(out, i16p), w)
        return out

    def dequant(self, bd, qp, coeff):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        self._dequant(bd, qp, w, h, ptr(coeff, i16p), self._s(coeff),
                      ptr(out, i16p), w)
        return out

    def quant_fast(self, bd, qp, intra_pic, coeff):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        n = self._quant_fast(bd, qp, intra_pic, w, h, ptr(coeff, i16p),
                             self._s(coeff), ptr(out, i16p), w)
        return out, n

    def quant_fast2(self, bd, qp, intra_pic, sign_hide, scan_order, coeff):
        h, w = coeff.shape
        out = np.zeros((h, w), np.int16)
        n = self._quant_fast2(bd, qp, intra_pic, sign_hide, scan_order, w, h,
                              ptr(coeff, i16p), self._s(coeff), ptr(out, i16p), self._s(out))
        return out, n

    def transform_matrix(self, tx, size):
        p = self._transform_matrix(tx, size)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(size, size)).copy()

    def min_max_mv(self, x, y, pw, ph, cx, cy, rng):
        mn = np.zeros(2, np.int32)
        mx = np.zeros(2, np.int32)
        self._min_max_mv(x, y, pw, ph, cx, cy, rng, ptr(mn, i32p),
                         ptr(mx, i32p))
        return mn, mx

    def tz_search(self, bd, blk, pw, ph, orig_pad, ref_pad, border):
        """orig_pad/ref_pad: padded luma planes; blk: MeBlock."""
        mv = np.zeros(2, np.int32)
        cost = C.c_uint32(0)
        o = orig_pad[border:, border:]
        r = ref_pad[border:, border:]
        self._tz_search(bd, C.byref(blk), pw, ph, ptr(o, u16p),
                        self._s(orig_pad), ptr(r, u16p), self._s(ref_pad),
                        ptr(mv, i32p), C.byref(cost))
        return (int(mv[0]), int(mv[1])), cost.value

    def subpel_search(self, bd, blk, pw, ph, orig_pad, ref_pad, border, fullpel):
        mv = np.zeros(2, np.int32)
        fp = np.array(fullpel, np.int32)
        dist = C.c_uint32(0)
        o = orig_pad[border:, border:]
        r = ref_pad[border:, border:]
        self._subpel_search(bd, C.byref(blk), pw, ph, ptr(o, u16p),
                            self._s(orig_pad), ptr(r, u16p), self._s(ref_pad),
                            ptr(fp, i32p), ptr(mv, i32p), C.byref(dist))
        return (int(mv[0]), int(mv[1])), dist.value

    def full_search(self, bd, x, y, w, h, fullpel_mv, mvp, lambda16, mn, mx,
                    target, ref_pad, border, pw, ph):
        mv = np.zeros(2, np.int32)
        mn = np.array(mn, np.int32)
        mx = np.array(mx, np.int32)
        r = ref_pad[border:, border:]
        if self.prefix == "xo":
            self._full_search(bd, x, y, w, h, fullpel_mv, mvp[0], mvp[1],
                              lambda16, ptr(mn, i32p), ptr(mx, i32p),
                              ptr(target, i16p), self._s(target), ptr(r, u16p),
                              self._s(ref_pad), ptr(mv, i32p))
        else:
            self._full_search(bd, x, y, w, h, fullpel_mv, mvp[0], mvp[1],
                              lambda16, ptr(mn, i32p), ptr(mx, i32p),
                              ptr(target, i16p), self._s(target), ptr(r, u16p),
                              self._s(ref_pad), pw, ph, ptr(mv, i32p))
        return int(mv[0]), int(mv[1])

    def mc_affine_block(self, bd, comp, x, y, w, h, mv3, pw, ph, padded, border):
        """mv3: three (x, y) corner MVs in 1/16 pel."""
        sh = 1 if comp else 0
        out = np.zeros((h >> sh, w >> sh), np.uint16)
        mv = np.ascontiguousarray(mv3, np.int32).reshape(3, 2)
        r = padded[border:, border:]
        self._mc_affine_block(bd, comp, x, y, w, h, ptr(mv, i32p), pw, ph,
                              ptr(r, u16p), self._s(padded), ptr(out, u16p), self._s(out))
        return out

    def mc_metric(self, bd, metric, qp, strength, x, y, w, h, mv, pw, ph, orig_pad,
                  ref_pad, border):
        o = orig_pad[border:, border:]
        r = ref_pad[border:, border:]
        return int(self._mc_metric(bd, metric, qp, strength, x, y, w, h, mv[0], mv[1],
                                   pw, ph, ptr(o, u16p), self._s(orig_pad),
                                   ptr(r, u16p), self._s(ref_pad)))

    def bipred_search(self, bd, job, pw, ph, orig_pad, other_pad, search_pad,
                      border):
        """job: BiBlock; padded luma planes.  Returns ((mvx, mvy), dist)."""
        res = MeResult()
        o = orig_pad[border:, border:]
        a = other_pad[border:, border:]
        b = search_pad[border:, border:]
        self._bipred_search(bd, C.byref(job), pw, ph, ptr(o, u16p),
                            self._s(orig_pad), ptr(a, u16p), self._s(other_pad),
                            ptr(b, u16p), self._s(search_pad), C.byref(res))
        return (res.mv_x, res.mv_y), res.subpel_dist

    def mc_bipred_block(self, bd, comp, x, y, w, h, mv0, mv1, pw, ph, pad0,
                        pad1, border):
        sh = 1 if comp else 0
        out = np.zeros((h >> sh, w >> sh), np.uint16)
        r0 = pad0[border:, border:]
        r1 = pad1[border:, border:]
        self._mc_bipred_block(bd, comp, x, y, w, h, mv0[0], mv0[1], mv1[0],
                              mv1[1], pw, ph, ptr(r0, u16p), self._s(pad0),
                              ptr(r1, u16p), self._s(pad1), ptr(out, u16p),
                              self._s(out))
        return out

    def deblock(self, bd, pw, ph, bipred, beta, tc, sub, cus, cu_map, planes,
                borders, l0=None, l1=None):
        """planes: 3 padded arrays (modified in place); cus: CU_DTYPE array."""
        pp = (u16p * 3)()
        ss = (pd * 3)()
        for c in range(3):
            o = planes[c][borders[c]:, borders[c]:]
            pp[c] = ptr(o, u16p)
            ss[c] = self._s(planes[c])
        cus = np.ascontiguousarray(cus)
        cup = cus.ctypes.data_as(C.POINTER(CuInfo))
        if self.prefix == "xo":
            cu_map = np.ascontiguousarray(cu_map, np.int32)
            self._deblock_picture(bd, pw, ph, bipred, beta, tc, sub, cup,
                                  ptr(cu_map, i32p), cu_map.shape[1], pp, ss)
        else:
            l0 = np.array(l0 if l0 is not None else [0], np.int32)
            l1 = np.array(l1 if l1 is not None else [0], np.int32)
            self._deblock_picture(bd, pw, ph, bipred, beta, tc, sub, cup,
                                  len(cus), pp, ss, ptr(l0, i32p), len(l0),
                                  ptr(l1, i32p), len(l1))

    def pad_border(self, w, h, planes, borders):
        """planes: 3 padded arrays 4:2:0 (modified in place)."""
        if self.prefix == "xo":
            for c in range(3):
                cw, ch = (w, h) if c == 0 else (w // 2, h // 2)
                o = planes[c][borders[c]:, borders[c]:]
                self._pad_border(cw, ch, borders[c], borders[c], ptr(o, u16p),
                                 self._s(planes[c]))
        else:
            pp = (u16p * 3)()
            ss = (pd * 3)()
            for c in range(3):
                o = planes[c][borders[c]:, borders[c]:]
                pp[c] = ptr(o, u16p)
                ss[c] = self._s(planes[c])
            self._pad_border(w, h, pp, ss)

    def residual_pipeline(self, bd, blk, orig, pred):
        """orig/pred: whole component planes (2-D); returns rec plane, coeff."""
        rec = pred.copy()
        coeff = np.zeros((blk.h, blk.w), np.int16)
        n = self._residual_pipeline(bd, C.byref(blk), ptr(orig, u16p),
                                    self._s(orig), ptr(pred, u16p),
                                    self._s(pred), ptr(rec, u16p), self._s(rec),
                                    ptr(coeff, i16p))
        return rec, coeff, n


def have_ref():
    return os.path.exists(REF_SO)
