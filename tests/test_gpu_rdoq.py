"""RDOQ on the device: RdoQuant::QuantRdo with
CoeffSignHideRdo through the C-ABI against (1) the golden vectors captured from
the reference build (tests/golden/rdoq.npz) and (2) the pinned oracle inside the
whole TransformAndReconstruct pipeline on random pictures, every block shape."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
import oracle_rdoq as oq
from helpers import rnd_samples

pytestmark = pytest.mark.gpu
G_BL = 128


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


@pytest.mark.parametrize("prove_zero", [0, 1])
def test_gpu_rdoq_golden(gpu, prove_zero):
    """Bit-exact levels and non-zero counts for the 432 reference vectors, with the
    all-zero proof ahead of the walk forced off and on."""
    import os
    api, ctx = gpu
    ctx.set_rdoq_prove_zero(prove_zero)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdoq.npz"))
    cases = g["cases"]
    done = 0
    for bd in (8, 10, 12):
        sel = [i for i, c in enumerate(cases) if c[0] == bd]
        # the 2-wide blocks of a rdo_quant_2x2 = 0 encoder take QuantFast: not this entry
        sel = [i for i in sel
               if not ((cases[i][5] == 2 or cases[i][6] == 2) and
                       (g["params"][i].view(oq.RDOQ_PARAMS_DTYPE)["flags"][0] & oq.RDOQ_NO_2X2))]
        blocks = np.zeros(len(sel), api.TX_DTYPE)
        params = np.zeros(len(sel), api.RDOQ_PARAMS_DTYPE)
        ctxs = np.zeros(len(sel), api.RDOQ_CTX_DTYPE)
        off, coeffs = [], []
        for k, i in enumerate(sel):
            _, cqp, comp, scan, sign_hide, w, h, nnz = cases[i]
            blocks[k] = (0, 0, w, h, comp, 0, 0, 0, cqp,
                         api.TXF_RDOQ | (0 if sign_hide else api.TXF_NO_SIGN_HIDING) |
                         (int(scan) << api.TXF_SCAN_SHIFT))
            params[k] = g["params"][i].view(api.RDOQ_PARAMS_DTYPE)[0]
            params[k]["ctx_index"] = k
            ctxs[k] = g["contexts"][i].view(api.RDOQ_CTX_DTYPE)[0]
            off.append(sum(len(c) for c in coeffs))
            coeffs.append(np.ascontiguousarray(g["src"][i][:h, :w]).reshape(-1))
        levels, nnz = ctx.quant_rdo_batch(bd, blocks, np.concatenate(coeffs),
                                          np.array(off, np.uint32), ctxs, params)
        for k, i in enumerate(sel):
            _, cqp, comp, scan, sign_hide, w, h, e_nnz = cases[i]
            got = levels[off[k]:off[k] + w * h].reshape(h, w)
            # nnz == 0 (cbf = 0): the reference leaves whatever its search wrote in
            # the level buffer and never reads it; the device writes zeros
            exp = g["levels"][i][:h, :w] if e_nnz else np.zeros((h, w), np.int16)
            assert nnz[k] == e_nnz and np.array_equal(got, exp), \
                (bd, i, w, h, comp, scan, sign_hide)
            done += 1
    ctx.set_rdoq_prove_zero(-1)
    assert done > 400


@pytest.mark.parametrize("bd", [8, 10])
def test_gpu_residual_rdoq_pipeline(gpu, bd):
    """xvcgpu_residual_rdoq_batch == the oracle's TransformAndReconstruct with
    QuantRdo: levels, counts and reconstruction; all block shapes 2..64 (both
    kernels: one wave per block up to 16x16, the workgroup path above / 2-wide),
    luma and chroma, a mix of RDOQ and QuantFast blocks in one batch."""
    api, ctx = gpu
    xo = ol.Lib("xo")
    rng = np.random.default_rng(930 + bd)
    pw, ph = 256, 192
    pad = lambda planes: [np.ascontiguousarray(np.pad(p, G_BL >> (1 if c else 0), mode="edge"))
                          for c, p in enumerate(planes)]
    base = [rnd_samples(rng, bd, ph >> (1 if c else 0), pw >> (1 if c else 0), 1) for c in range(3)]
    noise = [rng.integers(-24, 25, p.shape) << (bd - 8) for p in base]
    orig = pad([np.clip(p.astype(np.int64) + n, 0, (1 << bd) - 1).astype(np.uint16)
                for p, n in zip(base, noise)])
    pred = pad(base)
    O, P, R = (ctx.picture(pw, ph, bd) for _ in range(3))
    O.upload(orig, G_BL)
    P.upload(pred, G_BL)
    # a tiling of the luma plane into blocks of mixed shapes, and of the chroma planes
    sizes = [2, 4, 8, 16, 32, 64]
    blocks, params = [], []
    ctxs = np.concatenate([oq.random_contexts(rng) for _ in range(8)]).view(api.RDOQ_CTX_DTYPE)
    for comp in range(3):
        cw, chh = pw >> (1 if comp else 0), ph >> (1 if comp else 0)
        y = 0
        while y < chh:
            h = int(rng.choice([s for s in sizes if y % s == 0 and y + s <= chh and
                                (comp == 0 or s <= 32)]))
            x = 0
            while x < cw:
                w = int(rng.choice([s for s in sizes if x % s == 0 and x + s <= cw and
                                    (comp == 0 or s <= 32)]))
                intra = bool(rng.integers(0, 2))
                sc = 1 if comp else 0
                scan = int(rng.integers(0, 3)) if intra and (w << sc) < 16 and (h << sc) < 16 else 0
                rdoq = rng.integers(0, 5) != 0
                qp = int(rng.integers(20, 40))
                flags = (api.TXF_RDOQ if rdoq else 0) | (scan << api.TXF_SCAN_SHIFT) | \
                    (api.TXF_NO_SIGN_HIDING if rng.integers(0, 6) == 0 else 0)
                txh = txv = 0
                if comp == 0 and max(w, h) <= 64 and min(w, h) >= 4 and rng.integers(0, 3) == 0:
                    txh, txv = int(rng.choice([3, 5])), int(rng.choice([3, 5]))
                blocks.append((x, y, w, h, comp, txh, txv, 0, qp, flags))
                lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
                prm = np.zeros(1, api.RDOQ_PARAMS_DTYPE)
                prm["lambda"] = int(lam * 65536 + 0.5)
                inv_scale = [40, 45, 51, 57, 64, 72][(qp + 6 * (bd - 8)) % 6] << ((qp + 6 * (bd - 8)) // 6)
                prm["rd_factor"] = int(inv_scale * inv_scale / lam / 16 / (1 << (2 * (bd - 8))) + 0.5)
                prm["ctx_index"] = int(rng.integers(0, 8))
                prm["flags"] = (api.RDOQ_INTRA_CU if intra else 0) | \
                    (api.RDOQ_NO_2X2 if rng.integers(0, 3) == 0 else 0)
                params.append(prm[0])
                x += w
            y += h
    blocks = np.array(blocks, api.TX_DTYPE)
    params = np.array(params, api.RDOQ_PARAMS_DTYPE)
    levels, off, nnz = ctx.residual_rdoq_batch(O, P, R, blocks, ctxs, params)
    got = R.download(0)
    # oracle
    f = xo.dll.xo_residual_pipeline_rdoq
    f.restype = C.c_int
    vp = C.c_void_p
    exp = [np.zeros_like(p) for p in got]
    coeff = np.zeros(64 * 64, np.int16)
    n_coded = 0
    for i, b in enumerate(blocks):
        c = int(b["comp"])
        bb = G_BL >> (1 if c else 0)
        o, p = orig[c], pred[c]
        st = o.strides[0] // 2
        e_nnz = f(bd, vp(blocks[i:i + 1].ctypes.data), vp(ctxs.ctypes.data),
                  vp(params[i:i + 1].ctypes.data),
                  vp(o.ctypes.data + 2 * (bb * st + bb)), C.c_ssize_t(st),
                  vp(p.ctypes.data + 2 * (bb * st + bb)), C.c_ssize_t(st),
                  vp(exp[c].ctypes.data), C.c_ssize_t(exp[c].strides[0] // 2),
                  vp(coeff.ctypes.data))
        w, h = int(b["w"]), int(b["h"])
        assert nnz[i] == e_nnz, (i, tuple(b), int(nnz[i]), e_nnz)
        if e_nnz:       # cbf = 0: level buffer unspecified in the reference, zeros here
            assert np.array_equal(levels[off[i]:off[i] + w * h], coeff[:w * h]), (i, tuple(b))
        else:
            assert not levels[off[i]:off[i] + w * h].any(), (i, tuple(b))
        n_coded += e_nnz > 0
    for c in range(3):
        assert np.array_equal(got[c], exp[c]), c
    assert n_coded > len(blocks) // 3
    for p in (O, P, R):
        p.destroy()


def test_gpu_all_zero_proof_random_blocks(gpu):
    """Random blocks around the quantiser's threshold (the generator of
    tests/test_rdoq_zero_proof.py: sizes 4..32, three scan orders, random context
    states) through xvcgpu_quant_rdo_batch with the all-zero proof off and forced on:
    the same levels and counts, equal to the oracle's, and the proof takes blocks off
    the class lists."""
    import ctypes as C
    import test_rdoq_zero_proof as zp
    api, ctx = gpu
    xo = ol.Lib("xo")
    rng = np.random.default_rng(int(os.environ.get("XVC_SOAK", 0)) * 7919 + 4242)
    for bd in (8, 10, 12):
        cases = [c for c in zp._cases(rng, 4000) if c[0] == bd]
        snaps = np.concatenate([oq.random_contexts(rng) for _ in range(8)]).view(api.RDOQ_CTX_DTYPE)
        # blocks of one snapshot side by side (a workgroup of the proof shares one)
        order = np.argsort(rng.integers(0, 8, len(cases)), kind="stable")
        which = np.sort(rng.integers(0, 8, len(cases)))
        blocks = np.zeros(len(cases), api.TX_DTYPE)
        params = np.zeros(len(cases), api.RDOQ_PARAMS_DTYPE)
        off, coeffs, expect = [], [], []
        for k, j in enumerate(order):
            _, qp, comp, scan, _, prm, src = cases[j]
            h, w = src.shape
            blocks[k] = (0, 0, w, h, comp, 0, 0, 0, qp, api.TXF_RDOQ | (scan << api.TXF_SCAN_SHIFT))
            params[k] = prm[0]
            params[k]["ctx_index"] = which[k]
            off.append(sum(len(c) for c in coeffs))
            coeffs.append(src.reshape(-1))
            nnz, lv = oq.quant_rdo_oracle(xo, bd, qp, comp, scan, 1, snaps[which[k]:which[k] + 1],
                                          params[k:k + 1], src)
            expect.append((nnz, lv if nnz else np.zeros_like(lv)))
        listed = []
        for mode in (0, 1):
            ctx.set_rdoq_prove_zero(mode)
            levels, nnz = ctx.quant_rdo_batch(bd, blocks, np.concatenate(coeffs),
                                              np.array(off, np.uint32), snaps, params)
            cc = (C.c_int32 * 3)()
            ctx._check(ctx.lib.xvcgpu_quant_rdo_class_counts(ctx.h, cc))
            listed.append(sum(cc))
            for k, (e_nnz, e_lv) in enumerate(expect):
                h, w = e_lv.shape
                assert nnz[k] == e_nnz and np.array_equal(
                    levels[off[k]:off[k] + w * h].reshape(h, w), e_lv), (bd, mode, k, blocks[k])
        ctx.set_rdoq_prove_zero(-1)
        assert listed[1] < listed[0], listed


def test_four_lane_only_promise_is_checked(gpu):
    """xvcgpu_quant_rdo_set_four_lane_only leaves the general class's launch out; a block of
    that class in such a batch is reported by the next xvcgpu_sync instead of being dropped
    silently, and the same batch without the promise is quantised."""
    api, ctx = gpu
    rng = np.random.default_rng(5)
    bd, qp = 10, 30
    from xvc_amd import pipeline
    ctxs = pipeline.rdoq_init_contexts(qp, 1)
    lam, rdf = pipeline.rdoq_host_params(qp, bd)[0]
    shapes = [(16, 16), (8, 8), (64, 64)]          # the last one: general class
    blocks = np.zeros(len(shapes), api.TX_DTYPE)
    for i, (w, h) in enumerate(shapes):
        blocks[i]["w"], blocks[i]["h"], blocks[i]["qp"], blocks[i]["intra_pic"] = w, h, qp, api.TXF_RDOQ
    prm = np.zeros(len(shapes), api.RDOQ_PARAMS_DTYPE)
    prm["lambda"], prm["rd_factor"] = lam, rdf
    cf = [np.clip(np.rint(rng.laplace(0, 1, (h, w)) * 300), -32768, 32767).astype(np.int16).reshape(-1)
          for w, h in shapes]
    off = np.r_[0, np.cumsum([len(c) for c in cf])[:-1]].astype(np.uint32)
    want_lv, want_nnz = ctx.quant_rdo_batch(bd, blocks, np.concatenate(cf), off, ctxs, prm)
    assert (want_nnz > 0).all()
    ctx.set_rdoq_four_lane_only(True)
    try:
        with pytest.raises(api.XvcGpuError):
            ctx.quant_rdo_batch(bd, blocks, np.concatenate(cf), off, ctxs, prm)
        # the promise kept: the two four-lane blocks alone
        lv, nnz = ctx.quant_rdo_batch(bd, blocks[:2], np.concatenate(cf[:2]), off[:2], ctxs, prm[:2])
        assert np.array_equal(nnz, want_nnz[:2]) and np.array_equal(lv, want_lv[:len(lv)])
    finally:
        ctx.set_rdoq_four_lane_only(False)
    lv, nnz = ctx.quant_rdo_batch(bd, blocks, np.concatenate(cf), off, ctxs, prm)
    assert np.array_equal(nnz, want_nnz) and np.array_equal(lv, want_lv)
