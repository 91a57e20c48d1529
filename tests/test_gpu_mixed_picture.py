"""Decoder-side reconstruction of an inter picture with a mix of uni-pred,
bi-pred, affine, LIC and intra CUs (pipeline.MixedPictureDecoder: dependency waves over
the CU raster) against a CU-by-CU composition of the pinned oracle functions in
coding order, including deblocking and border extension."""
import ctypes as C

import numpy as np
import pytest

import oracle_intra as oi
import oracle_lib as ol
import oracle_lic
from helpers import rnd_samples

pytestmark = pytest.mark.gpu
BL, BC = 128, 64


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def oracle_picture(xo, api, pipeline, bd, qp, pw, ph, cu, orig, r0, r1, kind, mv0, mv1, imode,
                   mva):
    """orig: unpadded planes; r0 / r1: padded reference planes.  Raster order."""
    parts = pipeline.cu_partition(pw, ph, cu)
    qpc = pipeline.chroma_qp(qp)
    rec = [np.zeros_like(p) for p in orig]
    pred = [np.zeros_like(p) for p in orig]
    levels, nnz = [], []
    rp = xo.dll.xo_residual_pipeline
    rp.restype = C.c_int
    rp.argtypes = [C.c_int, C.c_void_p] + [oi.u16p, oi.pd] * 3 + [C.POINTER(C.c_int16)]
    ip = xo.dll.xo_intra_pred_block
    ip.restype = None
    ip.argtypes = [C.c_int, C.c_void_p, oi.u16p, oi.pd, oi.u16p, oi.pd]
    coeff = np.zeros(64 * 64, np.int16)
    borders = [BL, BC, BC]

    def u16(a):
        return C.cast(a.ctypes.data, oi.u16p), a.strides[0] // 2

    for i, (x, y, w, h) in enumerate(parts):
        for c in range(3):
            s = 1 if c else 0
            cx, cy, cw, ch = x >> s, y >> s, w >> s, h >> s
            if kind[i] == 0:
                p = xo.mc_block(bd, c, x, y, w, h, *mv0[i], pw, ph, r0[c], borders[c])
            elif kind[i] == 1:
                p = xo.mc_bipred_block(bd, c, x, y, w, h, mv0[i], mv1[i], pw, ph, r0[c], r1[c],
                                       borders[c])
            elif kind[i] == 4:
                p = xo.mc_affine_block(bd, c, x, y, w, h, mva[i], pw, ph, r0[c], borders[c])
            elif kind[i] == 2:
                j = np.zeros(1, oracle_lic.LIC_DTYPE)[0]
                j["x"], j["y"], j["w"], j["h"], j["comp"] = x, y, w, h, c
                j["neighbors"] = (1 if y else 0) | (2 if x else 0)
                j["mv_x"], j["mv_y"] = mv0[i]
                j["above_x"], j["above_y"] = x, max(0, y - cu)
                j["left_x"], j["left_y"] = max(0, x - cu), y
                full = oracle_lic.xo_mc_lic(xo, bd, j, pw, ph, r0, borders, rec)
                p = full[cy:cy + ch, cx:cx + cw]
            else:
                j = np.zeros(1, oi.INTRA_DTYPE)
                nb = (oi.HAS_LEFT if x else 0) | (oi.HAS_ABOVE if y else 0) | \
                    (oi.HAS_ABOVE_LEFT if x and y else 0)
                ar = max(0, min(h, pw - (x + w))) if y else 0
                j[0] = (cx, cy, cw, ch, c, imode[i], nb, ar >> s, 0, 0)
                ip(bd, j.ctypes.data, *u16(rec[c]), *u16(pred[c]))
                p = None
            if p is not None:
                pred[c][cy:cy + ch, cx:cx + cw] = p
            t = np.zeros(1, api.TX_DTYPE)
            t[0] = (cx, cy, cw, ch, c, 0, 0, 0, qpc if c else qp, 0)
            n = rp(bd, t.ctypes.data, *u16(orig[c]), *u16(pred[c]), *u16(rec[c]),
                   C.cast(coeff.ctypes.data, C.POINTER(C.c_int16)))
            levels.append(coeff[:cw * ch].copy())
            nnz.append(n)
    return rec, levels, np.array(nnz, np.int32)


@pytest.mark.parametrize("pw,ph,bd,qp,cu", [(256, 192, 10, 30, 16), (352, 288, 8, 27, 16),
                                            (136, 72, 10, 37, 16), (136, 72, 10, 27, 8),
                                            (200, 120, 12, 32, 8), (256, 192, 10, 32, 32),
                                            (512, 384, 10, 27, 64)])
def test_mixed_picture_decode(gpu, pw, ph, bd, qp, cu):
    from xvc_amd import pipeline
    api, ctx = gpu
    xo = ol.Lib("xo")
    rng = np.random.default_rng(1200 + pw + bd + cu)
    mx = (1 << bd) - 1

    def padded(planes):
        return [np.ascontiguousarray(np.pad(p, BL >> (c > 0), mode="edge"))
                for c, p in enumerate(planes)]

    base = [rnd_samples(rng, bd, hh, ww, True)
            for ww, hh in ((pw, ph), (pw // 2, ph // 2), (pw // 2, ph // 2))]
    orig = [np.clip(p.astype(np.int64) + rng.integers(-4, 5, p.shape) * (1 << (bd - 6)), 0, mx).astype(np.uint16)
            for p in base]
    r0 = padded([np.clip(p.astype(np.float64) * 0.9 + 12, 0, mx).astype(np.uint16) for p in base])
    r1 = padded([np.clip(p.astype(np.int64) + rng.integers(-6, 7, p.shape), 0, mx)
                 .astype(np.uint16) for p in base])
    parts = pipeline.cu_partition(pw, ph, cu)
    n = len(parts)
    kind = rng.choice(5, n, p=[0.35, 0.15, 0.2, 0.2, 0.1])
    kind[rng.choice(n, 5, replace=False)] = np.arange(5)     # every kind at least once
    mv0 = rng.integers(-70, 71, (n, 2))
    mva = mv0[:, None, :] + rng.integers(-12, 13, (n, 3, 2))   # corner vectors
    mv1 = rng.integers(-70, 71, (n, 2))
    imode = rng.integers(0, 67, n)
    e_rec, levels, nnz = oracle_picture(xo, api, pipeline, bd, qp, pw, ph, cu, orig, r0, r1,
                                        kind, mv0, mv1, imode, mva)
    assert np.count_nonzero(nnz) > 0 and len(set(kind.tolist())) == 5
    # in-loop filter + border on the oracle side
    cus = np.zeros(n, api.CU_DTYPE)
    cmap = -np.ones(((ph + 3) // 4, (pw + 3) // 4), np.int32)
    for i, (x, y, w, h) in enumerate(parts):
        c = cus[i]
        c["x"], c["y"], c["w"], c["h"] = x, y, w, h
        c["qp_y"], c["qp_c"] = qp, pipeline.chroma_qp(qp)
        c["intra"] = int(kind[i] == 3)
        c["cbf_luma"] = int(nnz[3 * i] != 0)
        c["ref_poc"][0] = -1 if kind[i] == 3 else 0
        c["ref_poc"][1] = 16 if kind[i] == 1 else -1
        if kind[i] == 4:    # corner vectors: top-left, top-right, bottom-left, bottom-right
            tl, tr, bl = mva[i]
            c["mv"][0][:] = [tl, tr, bl, tr + bl - tl]
        elif kind[i] != 3:
            c["mv"][0][:] = mv0[i]
        if kind[i] == 1:
            c["mv"][1][:] = mv1[i]
        cmap[y // 4:(y + h + 3) // 4, x // 4:(x + w + 3) // 4] = i
    e_pad = padded(e_rec)
    xo.deblock(bd, pw, ph, 1, 0, 0, 4, cus, cmap, e_pad, [BL, BC, BC], l0=[0], l1=[16])
    xo.pad_border(pw, ph, e_pad, [BL, BC, BC])
    # device
    R0, R1, D = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R0.upload(r0, BL)
    R1.upload(r1, BL)
    dec = pipeline.MixedPictureDecoder(ctx, pw, ph, bd, qp, kind, mv0, mv1, imode, cu,
                                       mv_affine=mva)
    assert max(g[0] for g in dec.groups) >= (3 if cu <= 16 else 1)   # real dependency chains
    dec.load(levels, nnz)
    dec.decode(R0, R1, D)
    ctx.sync()
    got = D.download()
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    ctx.deblock(D, cus, cmap, bipred=1)
    ctx.pad_border(D)
    ctx.sync()
    got = D.download(BL)
    for c in range(3):
        assert np.array_equal(got[c], e_pad[c]), c
    dec.destroy()
    for p in (R0, R1, D):
        p.destroy()
