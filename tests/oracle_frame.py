"""ctypes binding of the oracle's frame pass (oracle/xvc_oracle_frame.c).
TEST / CPU-BASELINE INFRASTRUCTURE ONLY: imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(), never by xvc_amd."""
import ctypes as C

import numpy as np

import oracle_lib as ol

u16p = C.POINTER(C.c_uint16)


class FrameArgs(C.Structure):
    _fields_ = [
        ("bd", C.c_int), ("pic_w", C.c_int), ("pic_h", C.c_int),
        ("n_cus", C.c_int), ("me_blocks", C.c_void_p),
        ("n_tx", C.c_int), ("tx_blocks", C.c_void_p),
        ("luma_tx_index", C.c_void_p), ("cu_map", C.c_void_p),
        ("map_stride", C.c_int),
        ("qp_y", C.c_int), ("qp_c", C.c_int), ("ref_poc", C.c_int),
        ("beta_offset", C.c_int), ("tc_offset", C.c_int), ("subblock", C.c_int),
        ("border", C.c_int * 3),
        ("orig", C.c_void_p * 3), ("orig_stride", C.c_ssize_t * 3),
        ("ref", C.c_void_p * 3), ("ref_stride", C.c_ssize_t * 3),
        ("pred", C.c_void_p * 3), ("pred_stride", C.c_ssize_t * 3),
        ("rec", C.c_void_p * 3), ("rec_stride", C.c_ssize_t * 3),
        ("me_results", C.c_void_p), ("nnz", C.c_void_p), ("cus", C.c_void_p),
        ("cu_base", C.c_int), ("encode_only", C.c_int),
        ("ssd", C.c_uint64 * 2),
        ("threads", C.c_int),
        ("rdoq_contexts", C.c_void_p), ("rdoq_params", C.c_void_p),
        ("rdoq_lambda", C.c_double),
    ]


def frame_pass(desc, bd, orig, ref, border, ref_poc=0, lib=None, encode_only=False,
               threads=1, reference=False):
    """desc: xvc_amd.pipeline.FrameDescriptors; orig/ref: [Y,U,V] padded uint16
    planes with `border` (luma) / border//2 (chroma) samples on each side.
    Returns (rec padded planes, me_results, nnz, cus, (ssd, samples))."""
    if reference:   # the same composition run by the reference's own classes
        lib = lib or ol.Lib("xr")
        f = lib.dll.xr_frame_pass
    else:
        lib = lib or ol.Lib("xo")
        f = lib.dll.xo_frame_pass
    f.restype = None
    f.argtypes = [C.POINTER(FrameArgs)]
    a = FrameArgs()
    a.bd, a.pic_w, a.pic_h = bd, desc.w, desc.h
    me = np.ascontiguousarray(desc.me)
    tx = np.ascontiguousarray(desc.tx)
    li = np.ascontiguousarray(desc.luma_idx)
    cm = np.ascontiguousarray(desc.cu_map, np.int32)
    a.n_cus, a.me_blocks = len(me), me.ctypes.data
    a.n_tx, a.tx_blocks = len(tx), tx.ctypes.data
    a.luma_tx_index, a.cu_map, a.map_stride = li.ctypes.data, cm.ctypes.data, cm.shape[1]
    a.qp_y, a.qp_c, a.ref_poc = desc.qp, desc.qp_c, ref_poc
    a.beta_offset, a.tc_offset, a.subblock = 0, 0, 4
    pred = [np.zeros_like(p) for p in ref]
    rec = [np.zeros_like(p) for p in ref]
    keep = []
    for c in range(3):
        b = border if c == 0 else border // 2
        a.border[c] = b
        for name, arr in (("orig", orig[c]), ("ref", ref[c]), ("pred", pred[c]),
                          ("rec", rec[c])):
            assert arr.dtype == np.uint16 and arr.flags["C_CONTIGUOUS"]
            getattr(a, name)[c] = arr.ctypes.data + (b * arr.strides[0] + b * 2)
            getattr(a, name + "_stride")[c] = arr.strides[0] // 2
            keep.append(arr)
    res = np.zeros(len(me), ol.MERES_DTYPE)
    nnz = np.zeros(len(tx), np.int32)
    cus = np.zeros(desc.n_cus_total, ol.CU_DTYPE)
    a.me_results, a.nnz, a.cus = res.ctypes.data, nnz.ctypes.data, cus.ctypes.data
    a.cu_base = desc.cu_base
    a.encode_only = 1 if encode_only else 0
    a.threads = threads
    if getattr(desc, "rdoq", False):    # RdoQuant::QuantRdo with the descriptors' inputs
        rc = np.ascontiguousarray(desc.rdoq_contexts)
        rp = np.ascontiguousarray(desc.rdoq_params)
        a.rdoq_contexts, a.rdoq_params = rc.ctypes.data, rp.ctypes.data
        a.rdoq_lambda = desc.rdoq_lambda
    f(C.byref(a))
    return rec, res, nnz, cus, (int(a.ssd[0]), int(a.ssd[1]))
