"""Whole-stream fixtures: the parsed syntax of every picture of a stream the
reference encoder produced (captured from the reference decoder by
oracle/ref_stream.cc through tools/gen_stream_golden.py) and what the decoder's
reconstruction stage must produce from it.

TEST INFRASTRUCTURE.  A fixture (tests/golden/stream_<name>.npz) holds
  stream           the bitstream (xvcenc container: 4-byte LE size + NAL) - data
  info             one STREAM_INFO record per picture, decoding order
  cus_<i>          STREAM_CU records of picture i, decoding (coding) order
  levels_<i>       int16 quantised levels, w*h per (CU, comp) with cbf
  post_<i>_<c>     final planes (optional: small clips only; always the MD5)
  pre_<i>_<c>      planes before the in-loop filter (optional)
"""
import ctypes as C
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STREAM_CU_DTYPE = np.dtype([
    ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("tree", "u1"), ("pred_mode", "u1"),
    ("qp", "i1", (3,)), ("root_cbf", "u1"), ("cbf", "u1", (3,)), ("tx_skip", "u1", (3,)),
    ("dc_only", "u1", (3,)), ("tx_select_idx", "i1"), ("tx_type", "u1", (3, 2)),
    ("intra_mode", "i1", (3,)), ("intra_chroma_raw", "i1"), ("inter_dir", "u1"),
    ("skip", "u1"), ("merge", "u1"), ("affine", "u1"), ("lic", "u1"), ("fullpel", "u1"),
    ("ref_idx", "i1", (2,)), ("nb_flags", "u1", (3,)), ("nb_above_right", "u1", (3,)),
    ("nb_below_left", "u1", (3,)), ("depth", "u1"), ("ref_poc", "<i4", (2,)),
    ("mv", "<i4", (2, 4, 2)), ("level_off", "<u4", (3,))], align=True)

STREAM_INFO_DTYPE = np.dtype([
    ("poc", "<i4"), ("doc", "<i4"), ("tid", "<i4"), ("nal_type", "<i4"), ("pic_type", "<i4"),
    ("pic_qp", "<i4"), ("deblock", "<i4"), ("beta_offset", "<i4"), ("tc_offset", "<i4"),
    ("allow_lic", "<i4"), ("adaptive_qp", "<i4"), ("highest_layer", "<i4"), ("padded", "<i4"),
    ("width", "<i4"), ("height", "<i4"), ("bitdepth", "<i4"), ("two_trees", "<i4"),
    ("num_ref", "<i4", (2,)), ("ref_poc", "<i4", (2, 5)), ("n_cus", "<i4"), ("n_levels", "<i4"),
    ("md5", "u1", (16,)), ("conforming", "<i4")], align=True)


def picture_md5(planes, bitdepth):
    """Checksum::CalculateMd5, kMinOverhead (checksum.cc:95-138): one MD5 over
    the rows of Y, U, V; samples as bytes at 8 bit, little-endian 16 bit above."""
    m = hashlib.md5()
    for p in planes:
        a = np.ascontiguousarray(p)
        m.update((a.astype(np.uint8) if bitdepth == 8 else a.astype("<u2")).tobytes())
    return np.frombuffer(m.digest(), np.uint8)


class StreamFixture:
    def __init__(self, name):
        self.path = os.path.join(GOLDEN, "stream_%s.npz" % name)
        z = np.load(self.path)
        self.z = z
        self.stream = z["stream"]
        self.info = z["info"].view(STREAM_INFO_DTYPE).reshape(-1)
        self.n = len(self.info)

    def cus(self, i):
        return self.z["cus_%d" % i].view(STREAM_CU_DTYPE).reshape(-1)

    def levels(self, i):
        return self.z["levels_%d" % i]

    def has_planes(self, i, which="post"):
        return "%s_%d_0" % (which, i) in self.z.files

    def planes(self, i, which="post"):
        return [self.z["%s_%d_%d" % (which, i, c)] for c in range(3)]


def decode_with_reference(stream, keep_planes=True):
    """Run the reference decoder (oracle/_ref) on a stream; returns
    (info[], cus[], levels[], pre[], post[]) per picture in decoding order."""
    import oracle_lib as ol
    lib = C.CDLL(ol.REF_SO)
    assert lib.xr_stream_cu_size() == STREAM_CU_DTYPE.itemsize, \
        (lib.xr_stream_cu_size(), STREAM_CU_DTYPE.itemsize)
    assert lib.xr_stream_info_size() == STREAM_INFO_DTYPE.itemsize, \
        (lib.xr_stream_info_size(), STREAM_INFO_DTYPE.itemsize)
    buf = np.ascontiguousarray(stream, np.uint8)
    lib.xr_stream_decode.argtypes = [C.c_void_p, C.c_long, C.c_int]
    n = lib.xr_stream_decode(buf.ctypes.data, len(buf), 1 if keep_planes else 0)
    assert n > 0, "reference decoder reported %d" % n
    out = []
    for i in range(n):
        info = np.zeros(1, STREAM_INFO_DTYPE)
        lib.xr_stream_get_info(i, C.c_void_p(info.ctypes.data))
        info = info[0]
        cus = np.zeros(int(info["n_cus"]), STREAM_CU_DTYPE)
        lib.xr_stream_get_cus(i, C.c_void_p(cus.ctypes.data))
        lv = np.zeros(max(1, int(info["n_levels"])), np.int16)
        lib.xr_stream_get_levels(i, C.c_void_p(lv.ctypes.data))
        lv = lv[:int(info["n_levels"])]
        pre = post = None
        if keep_planes:
            w, h = int(info["width"]), int(info["height"])
            pre, post = [], []
            for which, dst in ((0, pre), (1, post)):
                for c in range(3):
                    a = np.zeros((h >> (1 if c else 0), w >> (1 if c else 0)), np.uint16)
                    lib.xr_stream_get_plane(i, which, c, C.c_void_p(a.ctypes.data))
                    dst.append(a)
        out.append((info, cus, lv, pre, post))
    lib.xr_stream_release()
    return out


# ---- the reconstruction stage's input format (include/xvc_syntax.h) ------------
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

CU_AFFINE, CU_LIC = 1, 2


def to_syntax(info, cus):
    """Fixture records -> (xvc_picture_syntax, xvc_cu_syntax[]): what a parser
    hands to the reconstruction stage (the capture-only fields are dropped)."""
    ps = np.zeros(1, PICTURE_SYNTAX_DTYPE)
    for k in ("width", "height", "bitdepth", "poc", "pic_type", "deblock", "beta_offset",
              "tc_offset", "num_ref", "ref_poc", "n_cus", "n_levels"):
        ps[k] = info[k]
    ps["pad_border"] = info["padded"]
    out = np.zeros(len(cus), CU_SYNTAX_DTYPE)
    for k in ("x", "y", "w", "h", "tree", "pred_mode", "qp", "inter_dir", "cbf", "tx_skip",
              "tx_type", "ref_idx", "intra_mode", "level_off"):
        out[k] = cus[k]
    out["flags"] = cus["affine"] * CU_AFFINE + cus["lic"] * CU_LIC
    out["mv"] = cus["mv"][:, :, :3]
    return ps, out


class PaddedPicture:
    """Host-side 4:2:0 picture with a replicated-border margin; planes[c] is the
    visible view, ptr(c) the address of sample (0,0), strides in samples."""

    def __init__(self, w, h, border=128):
        self.w, self.h, self.border = w, h, border
        self.full = []
        for c in range(3):
            s, b = (1 if c else 0), border >> (1 if c else 0)
            self.full.append(np.zeros(((h >> s) + 2 * b, (w >> s) + 2 * b), np.uint16))
        self.planes = [f[(border >> (1 if c else 0)):-(border >> (1 if c else 0)),
                         (border >> (1 if c else 0)):-(border >> (1 if c else 0))]
                       for c, f in enumerate(self.full)]

    def stride(self, c):
        return self.full[c].shape[1]

    def ptr(self, c):
        b = self.border >> (1 if c else 0)
        return self.full[c].ctypes.data + 2 * (b * self.stride(c) + b)


def oracle_decode_stream(pictures, check=None):
    """Decode pictures [(info, cus, levels), ...] (decoding order) with the
    oracle's xo_decode_picture.  Returns per picture (PaddedPicture, pre planes,
    nb) ; `check(i, pic, pre, nb)` is called after each picture if given."""
    import oracle_lib as ol
    lib = C.CDLL(ol.ORACLE_SO)
    done = {}
    results = []
    for i, (info, cus, lv) in enumerate(pictures):
        ps, cs = to_syntax(info, cus)
        w, h = int(info["width"]), int(info["height"])
        pic = PaddedPicture(w, h)
        slots, ref_slot = [], np.full((2, 5), -1, np.int32)
        for l in range(2):
            for k in range(int(info["num_ref"][l])):
                poc = int(info["ref_poc"][l][k])
                if done[poc] not in slots:
                    slots.append(done[poc])
                ref_slot[l, k] = slots.index(done[poc])
        ref_ptrs = (C.c_void_p * max(1, 3 * len(slots)))()
        for s, rp in enumerate(slots):
            for c in range(3):
                ref_ptrs[3 * s + c] = rp.ptr(c)
        planes = (C.c_void_p * 3)(*[pic.ptr(c) for c in range(3)])
        strides = (C.c_ssize_t * 3)(*[pic.stride(c) for c in range(3)])
        pre = [np.zeros((h >> (1 if c else 0), w >> (1 if c else 0)), np.uint16) for c in range(3)]
        pre_ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in pre])
        nb = np.zeros((len(cs), 9), np.uint8)
        lvc = np.ascontiguousarray(lv if len(lv) else np.zeros(1, np.int16), np.int16)
        lib.xo_decode_picture(C.c_void_p(ps.ctypes.data), C.c_void_p(cs.ctypes.data),
                              C.c_void_p(lvc.ctypes.data), ref_ptrs,
                              C.c_void_p(ref_slot.ctypes.data), planes, strides,
                              C.c_int(pic.border), C.c_void_p(nb.ctypes.data), pre_ptrs)
        done[int(info["poc"])] = pic
        results.append((pic, pre, nb))
        if check:
            check(i, pic, pre, nb)
    return results


def deblock_metadata(info, cus):
    """What the in-loop filter reads of a single-tree picture (deblocking_filter.cc:
    79-241) from the fixture's leaf CUs: xvcgpu_cu_info records (api.CU_DTYPE layout)
    and the 4x4 cell map - the arrays xvcgpu_deblock / xo_deblock_rows take."""
    from xvc_amd import api
    assert not int(info["two_trees"])
    w, h = int(info["width"]), int(info["height"])
    n = len(cus)
    out = np.zeros(n, api.CU_DTYPE)
    out["x"], out["y"], out["w"], out["h"] = cus["x"], cus["y"], cus["w"], cus["h"]
    out["intra"] = cus["pred_mode"] == 0
    out["cbf_luma"] = cus["cbf"][:, 0]
    out["qp_y"], out["qp_c"] = cus["qp"][:, 0], cus["qp"][:, 1]
    out["ref_idx0"] = cus["ref_idx"][:, 0]
    out["ref_poc"] = cus["ref_poc"]
    out["mv"] = cus["mv"]
    cu_map = np.full(((h + 3) // 4, (w + 3) // 4), -1, np.int32)
    for i in range(n):
        x, y = int(cus["x"][i]) // 4, int(cus["y"][i]) // 4
        cu_map[y:y + int(cus["h"][i]) // 4, x:x + int(cus["w"][i]) // 4] = i
    return out, cu_map
