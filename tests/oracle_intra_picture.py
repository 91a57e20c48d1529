"""CPU twin of xvc_amd.pipeline.IntraPicturePass built from the pinned oracle
functions: the same all-intra composition (raster CU order, SATD-minimal mode,
DM chroma, QuantFast residual) CU by CU.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

import oracle_intra as oi
import oracle_lib as ol


def run(xo, desc, bd, orig, modes=None):
    """orig: [Y,U,V] unpadded planes.  desc: pipeline.IntraPictureDescriptors
    (its mode fields are overwritten).  Walks the CUs in desc order (wave
    order - equivalent to raster order for the data dependencies).  Returns
    (rec planes, modes, levels, nnz)."""
    rec = [np.zeros_like(p) for p in orig]
    pred = [np.zeros_like(p) for p in orig]
    levels = np.zeros(desc.level_total, np.int16)
    nnz = np.zeros(len(desc.tx), np.int32)
    out_modes = np.zeros(desc.n_cus, np.int32)
    xo.dll.xo_residual_pipeline.restype = C.c_int
    xo.dll.xo_residual_pipeline.argtypes = [C.c_int, C.c_void_p, oi.u16p, oi.pd, oi.u16p,
                                            oi.pd, oi.u16p, oi.pd, C.POINTER(C.c_int16)]
    f = xo.dll.xo_intra_pred_block
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p, oi.u16p, oi.pd, oi.u16p, oi.pd]
    rp = xo.dll.xo_residual_pipeline     # own prototype: numpy records as blocks
    i16p = C.POINTER(C.c_int16)
    tx = np.ascontiguousarray(desc.tx)
    coeff = np.zeros(64 * 64, np.int16)

    def u16(a):
        return C.cast(a.ctypes.data, oi.u16p), a.strides[0] // 2
    for i in range(desc.n_cus):
        j = desc.luma[i:i + 1]
        if modes is None:
            dist = oi.satd_modes(xo, "xo", bd, j[0], orig[0], rec[0])
            m = int(dist.argmin())
        else:
            m = int(modes[i])
        out_modes[i] = m
        desc.set_modes(i, i + 1, [m])
        for c, job in ((0, desc.luma[i:i + 1]), (1, desc.chroma[2 * i:2 * i + 1]),
                       (2, desc.chroma[2 * i + 1:2 * i + 2])):
            job = np.ascontiguousarray(job)
            f(bd, job.ctypes.data, C.cast(rec[c].ctypes.data, oi.u16p), rec[c].strides[0] // 2,
              C.cast(pred[c].ctypes.data, oi.u16p), pred[c].strides[0] // 2)
            k = 3 * i + c
            tx[k] = desc.tx[k]              # flags follow the chosen mode
            w, h = int(tx[k]["w"]), int(tx[k]["h"])
            n = rp(bd, C.c_void_p(tx.ctypes.data + k * tx.itemsize), *u16(orig[c]),
                   *u16(pred[c]), *u16(rec[c]), C.cast(coeff.ctypes.data, i16p))
            off = int(desc.level_off[k])
            levels[off:off + w * h] = coeff[:w * h]
            nnz[k] = n
    return rec, out_modes, levels, nnz
