"""ctypes helpers around the RDOQ oracle (oracle/xvc_oracle_rdoq.c) and, when
present, RdoQuant::QuantRdo of the reference build.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import oracle_lib as ol

RDOQ_CTX_DTYPE = np.dtype([
    ("csbf", "u1", (2, 2)), ("sig_luma", "u1", (54,)), ("sig_chroma", "u1", (12,)),
    ("greater1_luma", "u1", (16,)), ("greater1_chroma", "u1", (6,)),
    ("last_x_luma", "u1", (25,)), ("last_y_luma", "u1", (25,)), ("last_x_chroma", "u1", (3,)),
    ("last_y_chroma", "u1", (3,)), ("cbf_luma", "u1"), ("cbf_chroma", "u1"), ("root_cbf", "u1"),
    ("reserved", "u1")])
RDOQ_PARAMS_DTYPE = np.dtype([("lambda", "<i8"), ("rd_factor", "<i8"), ("ctx_index", "<u2"),
                              ("flags", "u1"), ("reserved", "u1", (5,))])
assert RDOQ_CTX_DTYPE.itemsize == 152 and RDOQ_PARAMS_DTYPE.itemsize == 24
RDOQ_INTRA_CU, RDOQ_NO_2X2 = 1, 2

_vp = C.c_void_p


def random_contexts(rng):
    """Any ContextModel state is legal input: (state << 1) | mps, 0..127."""
    c = np.zeros(1, RDOQ_CTX_DTYPE)
    c.view(np.uint8)[:] = rng.integers(0, 128, RDOQ_CTX_DTYPE.itemsize, dtype=np.uint8)
    c["reserved"] = 0
    return c


def init_contexts(xr, bd, qp, pic_type):
    c = np.zeros(1, RDOQ_CTX_DTYPE)
    xr.dll.xr_rdoq_init_contexts(bd, qp, pic_type, _vp(c.ctypes.data))
    return c


def quant_rdo_oracle(xo, bd, comp_qp, comp, scan_order, sign_hide, ctx, prm, src):
    h, w = src.shape
    out = np.zeros((h, w), np.int16)
    f = xo.dll.xo_quant_rdo
    f.restype = C.c_int
    s = np.ascontiguousarray(src, np.int16)
    nnz = f(bd, comp_qp, comp, scan_order, sign_hide, w, h, _vp(ctx.ctypes.data),
            _vp(prm.ctypes.data), _vp(s.ctypes.data), C.c_ssize_t(w), _vp(out.ctypes.data),
            C.c_ssize_t(w))
    return nnz, out


def quant_rdo_reference(xr, bd, qp_luma, lam, comp, scan_order, sign_hide, ctx, flags, src):
    """Returns (nnz, levels, params the reference derived, the component's raw qp)."""
    h, w = src.shape
    out = np.zeros((h, w), np.int16)
    prm = np.zeros(1, RDOQ_PARAMS_DTYPE)
    prm["flags"] = flags
    cqp = C.c_int(0)
    f = xr.dll.xr_quant_rdo
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_double] + [C.c_int] * 5 + [_vp, _vp, C.POINTER(C.c_int),
                                                                  _vp, C.c_ssize_t, _vp,
                                                                  C.c_ssize_t]
    s = np.ascontiguousarray(src, np.int16)
    nnz = f(bd, qp_luma, lam, comp, scan_order, sign_hide, w, h, ctx.ctypes.data,
            prm.ctypes.data, C.byref(cqp), s.ctypes.data, w, out.ctypes.data, w)
    return nnz, out, prm, cqp.value


def random_coeffs(rng, w, h, bd, kind, qp=24):
    """Transform-coefficient-like blocks: energy concentrated at low
    frequencies, a range of magnitudes incl. the int16 extremes."""
    yy, xx = np.mgrid[0:h, 0:w]
    decay = np.exp(-(xx / max(1.0, w / 3.0) + yy / max(1.0, h / 3.0)))
    # quantiser step ~ 2^(qp/6) at this transform gain: levels of about 1, 4, 40
    # and up to the int16 limits
    amp = [1.5, 6, 60, 30000 / 2.0 ** (qp / 6.0)][kind % 4] * 2.0 ** (qp / 6.0) * \
        (1 << (bd - 8)) / 2.0
    c = rng.laplace(0, 1, (h, w)) * amp * decay
    if kind >= 4:       # dense small values: exercises the zero-out decisions
        c = rng.laplace(0, 1, (h, w)) * amp / 8
    c = np.clip(np.rint(c), -32768, 32767).astype(np.int16)
    if w >= 64:
        c[:, 32:] = 0   # ForwardTransform keeps 32 low-frequency outputs (transform.cc:1458)
    if h >= 64:
        c[32:, :] = 0
    return c
