"""Python binding of the C++ host layer's picture decoder (libxvchost.so:
xvc_amd/host/xvc_picture_decoder.{h,cc}) - plumbing for tests and bench.py.

The decoder's reconstruction stage: parsed syntax of a picture
(include/xvc_syntax.h) in, reconstructed / filtered / padded device picture out.
All logic lives in the C++ class; this module only marshals arguments.  No CPU
fallback: without libxvchost.so / libxvcgpu.so / a gfx950 device it raises.
"""
import ctypes as C
import os

import numpy as np

from . import api

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(HERE, "libxvchost.so")

CU_SYNTAX_DTYPE = np.dtype([
    ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("tree", "u1"), ("pred_mode", "u1"),
    ("qp", "i1", (3,)), ("inter_dir", "u1"), ("cbf", "u1", (3,)), ("flags", "u1"),
    ("tx_skip", "u1", (3,)), ("reserved0", "u1"), ("tx_type", "u1", (3, 2)),
    ("ref_idx", "i1", (2,)), ("intra_mode", "i1", (3,)), ("reserved1", "u1"),
    ("mv", "<i4", (2, 3, 2)), ("level_off", "<u4", (3,))], align=True)
PICTURE_SYNTAX_DTYPE = np.dtype([
    ("width", "<i4"), ("height", "<i4"), ("bitdepth", "<i4"), ("poc", "<i4"),
    ("pic_type", "<i4"), ("deblock", "<i4"), ("beta_offset", "<i4"), ("tc_offset", "<i4"),
    ("pad_border", "<i4"), ("num_ref", "<i4", (2,)), ("ref_poc", "<i4", (2, 5)),
    ("n_cus", "<i4"), ("n_levels", "<i4")], align=True)
assert CU_SYNTAX_DTYPE.itemsize == 92 and PICTURE_SYNTAX_DTYPE.itemsize == 92

_host = None


def load_host_library():
    global _host
    if _host is not None:
        return _host
    api.load_library()      # libxvcgpu.so first (one HIP runtime per process)
    if not os.path.exists(HOST_LIB_PATH):
        raise api.XvcGpuError("libxvchost.so is not built: run `python -m xvc_amd.build`")
    lib = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    lib.xvc_host_picture_decoder_create.restype = vp
    lib.xvc_host_picture_decoder_create.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.xvc_host_picture_decoder_destroy.restype = None
    lib.xvc_host_picture_decoder_destroy.argtypes = [vp]
    lib.xvc_host_picture_decoder_decode.argtypes = [vp, vp, vp, vp, C.POINTER(vp), vp]
    lib.xvc_host_picture_decoder_decode_sequence.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.xvc_host_picture_decoder_waves.argtypes = [vp]
    lib.xvc_host_picture_decoder_launches.argtypes = [vp]
    lib.xvc_host_picture_decoder_one_launch_intra.argtypes = [vp, C.c_int]
    lib.xvc_host_picture_decoder_one_launch_intra.restype = None
    lib.xvc_host_plan_picture.argtypes = [vp, vp, vp, vp, vp]
    _host = lib
    return lib


def plan_picture(ps, cus, levels):
    """Host-only (no GPU): neighbour state per CU [n, 9] and wave per CU."""
    lib = load_host_library()
    n = len(cus)
    nb = np.zeros((n, 9), np.uint8)
    wave = np.zeros(n, np.int32)
    lv = np.ascontiguousarray(levels if len(levels) else np.zeros(1, np.int16), np.int16)
    n_waves = lib.xvc_host_plan_picture(ps.ctypes.data, cus.ctypes.data, lv.ctypes.data,
                                        nb.ctypes.data, wave.ctypes.data)
    if n_waves < 0:
        raise ValueError("malformed picture syntax (PictureDecoder::Validate)")
    return n_waves, nb, wave


class PictureDecoder:
    """xvc_gpu::PictureDecoder for one picture size on one context."""

    def __init__(self, ctx, width, height, bitdepth):
        self.ctx = ctx
        self.lib = load_host_library()
        self.h = self.lib.xvc_host_picture_decoder_create(ctx.h, width, height, bitdepth)
        if not self.h:
            raise api.XvcGpuError("xvc_host_picture_decoder_create failed")

    def decode(self, ps, cus, levels, ref_pics, rec):
        """ps: PICTURE_SYNTAX_DTYPE[1]; cus: CU_SYNTAX_DTYPE[n]; levels int16;
        ref_pics[list][idx]: api.Picture; rec: api.Picture (output)."""
        refs = (C.c_void_p * 10)()
        for l in range(2):
            for k, p in enumerate(ref_pics[l]):
                refs[l * 5 + k] = p.h_pic
        lv = np.ascontiguousarray(levels if len(levels) else np.zeros(1, np.int16), np.int16)
        st = self.lib.xvc_host_picture_decoder_decode(
            self.h, ps.ctypes.data, cus.ctypes.data, lv.ctypes.data, refs, rec.h_pic)
        self.ctx._check(st)

    def add_lane(self, ctx):
        """A further picture lane (PictureDecoder::AddLane): decode_sequence deals the
        pictures over the lanes, pictures that do not reference each other run side
        by side.  `ctx`: another api.Context on the same device (kept by the caller)."""
        self.lib.xvc_host_picture_decoder_add_lane.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx._check(self.lib.xvc_host_picture_decoder_add_lane(self.h, ctx.h))
        self._lanes = getattr(self, "_lanes", []) + [ctx]

    def decode_sequence(self, pictures, ref_index, recs):
        """pictures: [(ps, cus, levels)] in decoding order; ref_index[i][list][k]: the
        position in this sequence of picture i's reference (list, k), -1 = unused; recs:
        [api.Picture] outputs.  The C++ layer plans picture i + 1 on a worker thread
        while it uploads and launches picture i (xvc_gpu::PictureDecoder::DecodeSequence)."""
        n = len(pictures)
        keep = []
        ps_p, cu_p, lv_p = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, (ps, cus, levels) in enumerate(pictures):
            lv = np.ascontiguousarray(levels if len(levels) else np.zeros(1, np.int16), np.int16)
            keep.append(lv)
            ps_p[i], cu_p[i], lv_p[i] = ps.ctypes.data, cus.ctypes.data, lv.ctypes.data
        ri = np.ascontiguousarray(ref_index, np.int32).reshape(n, 2, 5)
        rp = (C.c_void_p * n)(*[r.h_pic for r in recs])
        st = self.lib.xvc_host_picture_decoder_decode_sequence(self.h, n, ps_p, cu_p, lv_p,
                                                               ri.ctypes.data, rp)
        self.ctx._check(st)

    @property
    def waves(self):
        return self.lib.xvc_host_picture_decoder_waves(self.h)

    @property
    def launches(self):
        return self.lib.xvc_host_picture_decoder_launches(self.h)

    def one_launch_intra(self, on):
        """Intra pictures: all dependency waves in one cooperative launch (default)
        or one launch set per wave."""
        self.lib.xvc_host_picture_decoder_one_launch_intra(self.h, int(bool(on)))

    def destroy(self):
        if self.h:
            self.lib.xvc_host_picture_decoder_destroy(self.h)
            self.h = None
