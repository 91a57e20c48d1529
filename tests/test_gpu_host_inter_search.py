"""The host control the C++ layer keeps around the motion-search batches
(xvc_gpu::InterSearch in xvc_amd/host/xvc_gpu_ops.h: EvalStartMvp,
EvalFinalMvpIdx, the bit prices, the per-list SearchRefIdx loop, the merge fold,
SearchBiIterative / SearchMotion) against the reference's own member functions
(oracle/_ref, xr_eval_start_mvp / xr_eval_final_mvp_idx / xr_mvd_bits /
xr_search_merge_candidates / xr_search_motion), on random CUs and predictors."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_parity import BL, make_pics, me_blocks, to_me_struct
from xvc_amd import decoder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def host_lib():
    L = decoder.load_host_library()
    L.xvc_host_eval_start_mvp_batch.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3
    L.xvc_host_eval_final_mvp_idx.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.xvc_host_mvd_bits.restype = C.c_uint32
    L.xvc_host_search_ref_idx_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
    return L


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
def test_bit_prices_and_final_mvp_vs_reference():
    L, xr = host_lib(), ol.Lib("xr").dll
    xr.xr_mvd_bits.restype = C.c_uint32
    xr.xr_eval_final_mvp_idx.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(11)
    for _ in range(3000):
        a = [int(v) for v in rng.integers(-3000, 3000, 4)]
        sh = int(rng.integers(0, 2)) * 2
        assert L.xvc_host_mvd_bits(*a, sh) == xr.xr_mvd_bits(*a, sh)
        mvp = np.array(rng.integers(-800, 800, 4), np.int32)
        if rng.integers(0, 4) == 0:
            mvp[2:] = mvp[:2]           # identical predictors: the start index breaks the tie
        mv = [int(v) for v in (mvp[:2] if rng.integers(0, 3) == 0 else rng.integers(-800, 800, 2))]
        fp, start = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        got = L.xvc_host_eval_final_mvp_idx(mvp.ctypes.data, mv[0], mv[1], start, fp)
        exp = xr.xr_eval_final_mvp_idx(fp, mvp.ctypes.data, mv[0], mv[1], start)
        assert got == exp, (mvp, mv, fp, start)


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("bd", [8, 10])
def test_eval_start_mvp_and_ref_loop_vs_reference(gpu, bd):
    api, ctx = gpu
    L, xr, xo = host_lib(), ol.Lib("xr"), ol.Lib("xo")
    xr.dll.xr_eval_start_mvp.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                         C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p,
                                         C.c_void_p]
    rng = np.random.default_rng(4200 + bd)
    pw, ph = 256, 160
    orig, ref0 = make_pics(rng, bd, pw, ph, BL, (5, -3))
    _, ref1 = make_pics(rng, bd, pw, ph, BL, (-9, 4))
    ref1 = np.ascontiguousarray(np.roll(ref0, (3, -7), (0, 1)))   # a second, shifted reference
    O, R0, R1 = (ctx.picture(pw, ph, bd) for _ in range(3))
    O.upload([orig, None, None], BL)
    R0.upload([ref0, None, None], BL)
    R1.upload([ref1, None, None], BL)
    n = 60
    blocks = me_blocks(rng, api, pw, ph, n)
    mvp = np.array(rng.integers(-160, 160, (2, n, 4)), np.int32)
    mvp[:, ::5, 2:] = mvp[:, ::5, :2]
    side = np.array(rng.integers(1, 6, (2, n)), np.uint32)
    idx, cost = np.zeros(n, np.int32), np.zeros(n, np.uint32)
    b = np.ascontiguousarray(blocks)
    refs = [(R0, ref0), (R1, ref1)]
    start = []
    for r, (R, ref) in enumerate(refs):
        m = np.ascontiguousarray(mvp[r])
        st = L.xvc_host_eval_start_mvp_batch(ctx.h, O.h_pic, R.h_pic, b.ctypes.data, n,
                                             m.ctypes.data, idx.ctypes.data, cost.ctypes.data)
        assert st == 0
        o, rr = orig[BL:, BL:], ref[BL:, BL:]
        for i in range(n):
            s = to_me_struct(blocks[i])
            c = C.c_uint32(0)
            e = xr.dll.xr_eval_start_mvp(bd, C.byref(s), pw, ph, o.ctypes.data, orig.strides[0] // 2,
                                         rr.ctypes.data, ref.strides[0] // 2, m[i].ctypes.data,
                                         C.byref(c))
            assert (int(idx[i]), int(cost[i])) == (e, c.value), (r, i, tuple(blocks[i]), m[i])
        start.append(idx.copy())
    # the per-list loop: its parts are pinned above / in test_gpu_parity (search); the fold
    # here against the same steps composed in Python from the reference's functions
    out = np.zeros((n, 6), np.int32)
    handles = (C.c_void_p * 2)(R0.h_pic, R1.h_pic)
    mv_flat, side_flat = np.ascontiguousarray(mvp), np.ascontiguousarray(side)
    assert L.xvc_host_search_ref_idx_batch(ctx.h, O.h_pic, handles, 2, b.ctypes.data, n,
                                           mv_flat.ctypes.data, side_flat.ctypes.data,
                                           out.ctypes.data) == 0
    xr.dll.xr_mvd_bits.restype = C.c_uint32
    xr.dll.xr_eval_final_mvp_idx.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    for i in range(n):
        best = None
        for r, (R, ref) in enumerate(refs):
            blk = blocks[i].copy()
            k = int(start[r][i])
            blk["mvp_x"], blk["mvp_y"] = mvp[r][i][2 * k], mvp[r][i][2 * k + 1]
            s = to_me_struct(blk)
            (fx, fy), _ = xo.tz_search(bd, s, pw, ph, orig, ref, BL)
            if blk["fullpel_mv"]:
                mv = (16 * fx, 16 * fy)
                dist = xo.mc_metric(bd, 1, 32, 16, s.x, s.y, s.w, s.h, mv, pw, ph, orig, ref, BL)
            else:
                mv, dist = xo.subpel_search(bd, s, pw, ph, orig, ref, BL, (fx, fy))
            fp = int(blk["fullpel_mv"])
            m = np.ascontiguousarray(mvp[r][i])
            fidx = xr.dll.xr_eval_final_mvp_idx(fp, m.ctypes.data, mv[0], mv[1], k)
            bits = int(side[r][i]) + 1 + xr.dll.xr_mvd_bits(int(m[2 * fidx]), int(m[2 * fidx + 1]),
                                                             mv[0], mv[1], 2 * fp)
            c = dist + ((bits * int(blk["lambda16"])) >> 16)
            if best is None or c < best[5]:
                best = (r, fidx, mv[0], mv[1], dist, c)
        assert tuple(int(v) for v in out[i]) == best, (i, tuple(blocks[i]))
    for p in (O, R0, R1):
        p.destroy()


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("bd", [8, 10])
def test_search_merge_candidates_vs_reference(gpu, bd):
    """SearchMergeCandidates (inter_search.cc:165-197): per-candidate SATD on the
    device (uni and bi-directional candidates), the double-precision fold, the
    stable sort and the 1.25x cut on the host - against the reference's member
    function on the same candidate lists."""
    api, ctx = gpu
    L, xr = host_lib(), ol.Lib("xr")
    L.xvc_host_search_merge_candidates_batch.argtypes = [C.c_void_p] * 6 + [C.c_int] + \
        [C.c_void_p] * 3
    f = xr.dll.xr_search_merge_candidates
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p,
                                  C.c_ssize_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5100 + bd)
    pw, ph = 256, 160
    orig, ref0 = make_pics(rng, bd, pw, ph, BL, (4, -2))
    ref1 = np.ascontiguousarray(np.roll(ref0, (2, 5), (0, 1)))
    O, R0, R1, P = (ctx.picture(pw, ph, bd) for _ in range(4))
    O.upload([orig, None, None], BL)
    R0.upload([ref0, None, None], BL)
    R1.upload([ref1, None, None], BL)
    # non-overlapping CUs on a 64-grid, all sizes
    blocks, n = [], 0
    for gy in range(0, ph - 63, 64):
        for gx in range(0, pw, 64):
            w = int(rng.choice([8, 16, 32, 64]))
            h = int(rng.choice([8, 16, 32, 64]))
            blocks.append((gx, gy, w, h))
    n = len(blocks)
    b = np.zeros(n, api.ME_DTYPE)
    for i, (x, y, w, h) in enumerate(blocks):
        b[i]["x"], b[i]["y"], b[i]["w"], b[i]["h"] = x, y, w, h
    K = 5                                    # constants::kNumInterMergeCandidates
    cands = np.zeros((n, K, 5), np.int32)
    cands[:, :, 0] = rng.integers(0, 3, (n, K))
    cands[:, :, 1:] = rng.integers(-96, 96, (n, K, 4))
    cands[::3, 2] = cands[::3, 1]            # equal candidates: the stable sort decides
    lam = np.array(rng.choice([2.0, 7.6, 19.3, 60.0], n), np.float64)
    out = np.zeros((n, K + 1), np.int32)
    assert L.xvc_host_search_merge_candidates_batch(
        ctx.h, O.h_pic, R0.h_pic, R1.h_pic, P.h_pic, b.ctypes.data, n, cands.ctypes.data,
        lam.ctypes.data, out.ctypes.data) == 0
    o, r0, r1 = orig[BL:, BL:], ref0[BL:, BL:], ref1[BL:, BL:]
    nums = set()
    for i, (x, y, w, h) in enumerate(blocks):
        exp = np.zeros(K, np.int32)
        c = np.ascontiguousarray(cands[i])
        num = f(bd, x, y, w, h, pw, ph, o.ctypes.data, orig.strides[0] // 2, r0.ctypes.data,
                ref0.strides[0] // 2, r1.ctypes.data, ref1.strides[0] // 2, c.ctypes.data,
                float(lam[i]), exp.ctypes.data, None)
        assert out[i][K] == num, (i, blocks[i], out[i].tolist(), exp.tolist(), num)
        assert out[i][:K].tolist() == exp.tolist(), (i, blocks[i], out[i].tolist(), exp.tolist())
        nums.add(num)
    assert len(nums) >= 2
    for p in (O, R0, R1, P):
        p.destroy()
    assert [L.xvc_host_choose_uni_or_bi(*t) for t in
            [(5, 5, 5), (5, 4, 6), (4, 5, 6), (5, 5, 6), (7, 7, 6)]] == [0, 2, 1, 1, 0]


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("bd,iterations", [(8, 4), (10, 4), (10, 1)])
def test_search_motion_vs_reference(gpu, bd, iterations):
    """InterSearch::SearchMotion (inter_search.cc:198-259) as a whole - the
    uni-directional searches of both lists, SearchBiIterative's loop around the
    device steps (bootstrap vectors and predictors carried between iterations, the
    stop rule) and the final choice - against the reference's member function on
    the same CUs, neighbours (AMVP lists) and pictures."""
    api, ctx = gpu
    L, xr = host_lib(), ol.Lib("xr").dll
    L.xvc_host_search_motion_batch.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + \
        [C.c_int, C.c_void_p]
    xr.xr_search_motion.restype = None
    xr.xr_search_motion.argtypes = [C.c_int] * 6 + [C.c_uint32] + [C.c_int] * 3 + \
        [C.c_void_p, C.c_ssize_t] * 3 + [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9100 + bd + iterations)
    pw, ph = 256, 160
    _, ref0 = make_pics(rng, bd, pw, ph, BL, (0, 0))
    _, ref1 = make_pics(rng, bd, pw, ph, BL, (0, 0), noise=5)
    # the original: the mean of the two references displaced differently + noise, so
    # that bi-prediction wins for many CUs and the refinement moves both vectors
    a = np.roll(ref0, (2, -5), (0, 1)).astype(np.int32)
    b2 = np.roll(ref1, (-3, 6), (0, 1)).astype(np.int32)
    orig = np.clip((a + b2 + 1) // 2 + rng.integers(-2, 3, a.shape), 0,
                   (1 << bd) - 1).astype(np.uint16)
    O, R0, R1 = (ctx.picture(pw, ph, bd) for _ in range(3))
    O.upload([orig, None, None], BL)
    R0.upload([ref0, None, None], BL)
    R1.upload([ref1, None, None], BL)
    n = 40
    blocks = np.zeros((2, n), api.ME_DTYPE)
    mvp = np.zeros((2, n, 4), np.int32)
    exp = np.zeros((n, 26), np.int64)
    o, r0, r1 = orig[BL:, BL:], ref0[BL:, BL:], ref1[BL:, BL:]
    for i in range(n):
        w, h = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 32, 64]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        fp = int(rng.integers(0, 6) == 0)
        lam = int(rng.choice([120000, 498000, 1500000]))
        nb = np.array(rng.integers(-150, 150, 8), np.int32)
        if i % 4 == 0:
            nb[4:] = nb[:4]              # identical predictors
        xr.xr_search_motion(bd, x, y, w, h, fp, lam, iterations, pw, ph, o.ctypes.data,
                            orig.strides[0] // 2, r0.ctypes.data, ref0.strides[0] // 2,
                            r1.ctypes.data, ref1.strides[0] // 2, nb.ctypes.data,
                            exp[i].ctypes.data)
        for l in range(2):
            bk = blocks[l][i]
            bk["x"], bk["y"], bk["w"], bk["h"] = x, y, w, h
            bk["depth_nonzero"], bk["fullpel_mv"], bk["lambda16"] = 1, fp, lam
            bk["search_range"] = exp[i][8 + l]
            mvp[l][i] = exp[i][10 + 4 * l:14 + 4 * l]
    assert len({tuple(m) for m in mvp.reshape(-1, 4)}) > n // 2     # real AMVP lists
    side_uni = np.full((2, n), 3, np.uint32)    # fast_inter_pred_bits: 3 (bi picture), 5 (bi CU)
    side_bi = np.full(n, 5, np.uint32)
    out = np.zeros((n, 18), np.int64)
    bl, mv = np.ascontiguousarray(blocks), np.ascontiguousarray(mvp)
    assert L.xvc_host_search_motion_batch(ctx.h, O.h_pic, R0.h_pic, R1.h_pic, bl.ctypes.data, n,
                                          mv.ctypes.data, side_uni.ctypes.data,
                                          side_bi.ctypes.data, iterations, out.ctypes.data) == 0
    n_bi = 0
    for i in range(n):
        e = exp[i]
        d = int(e[1])
        want = [d, 0, 0, 0, 0, 0, 0, int(e[0])]
        got = [int(v) for v in out[i][:8]]
        # the uni-directional halves first: {cost, mv, mvp_idx} per list
        assert [int(v) for v in out[i][8:16]] == [int(v) for v in e[18:26]], \
            (i, tuple(blocks[0][i]), mvp[:, i])
        for l in range(2):
            if d == 2 or d == l:
                want[1 + 2 * l:3 + 2 * l] = [int(e[2 + 2 * l]), int(e[3 + 2 * l])]
                want[5 + l] = int(e[6 + l])
            else:                       # the unused list: the reference cleared it
                got[1 + 2 * l:3 + 2 * l] = [0, 0]
                got[5 + l] = 0
        if got != want:
            print("CU", i, "host", " ".join(str(int(v)) for v in out[i]))
            print("CU", i, "ref ", " ".join(str(int(v)) for v in e))
        assert got == want, (i, tuple(blocks[0][i]))
        n_bi += d == 2
    assert n_bi >= n // 4
    for p in (O, R0, R1):
        p.destroy()


# (list 0 / list 1 as indices into the picture set, the picture set's POCs, current POC)
_REF_CONFIGS = {
    # hierarchical B, two pictures per list, the lists name the same two pictures:
    # every list-1 entry reuses list 0's search (inter_search.cc:536-542)
    "b2_shared": ([0, 1], [1, 0], [0, 16], 8),
    # two per list, list 1 partly unique
    "b2_mixed": ([0, 1], [2, 1], [4, 0, 16], 8),
    # placebo: three per list (encoder_settings.cc:36), all of list 1 unique
    "b3_unique": ([0, 1, 2], [3, 4, 5], [6, 4, 0, 10, 12, 16], 8),
    # only back references: list 1 of a bi-directional CU carries no vector
    # difference (PictureData::DetermineForceBipredL1MvdZero)
    "back_only": ([0, 1], [0, 1], [12, 8], 16),
}


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("config,bd,iterations", [
    ("b2_shared", 10, 1), ("b2_mixed", 10, 1), ("b2_mixed", 8, 4), ("b3_unique", 10, 4),
    ("b3_unique", 8, 1), ("back_only", 10, 1), ("back_only", 8, 4)])
def test_search_motion_multi_ref_vs_reference(gpu, config, bd, iterations):
    """InterSearch::SearchMotion as the reference configures itself
    (default_num_ref_pics 2, placebo 3; lists that share pictures; back-only
    reference structures; closed-form bit prices): xvc_gpu::InterSearch::
    SearchMotionMultiBatch against the reference's member function on the same
    CUs, neighbours (real AMVP lists per reference picture) and pictures."""
    api, ctx = gpu
    L, xr = host_lib(), ol.Lib("xr").dll
    l0, l1, pocs, cur = _REF_CONFIGS[config]
    L.xvc_host_search_motion_multi_batch.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int,
                                                                         C.c_void_p, C.c_int,
                                                                         C.c_void_p, C.c_int,
                                                                         C.c_void_p]
    xr.xr_search_motion_multi.restype = None
    xr.xr_search_motion_multi.argtypes = [C.c_int] * 6 + [C.c_uint32] + [C.c_int] * 3 + \
        [C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9900 + bd + iterations + len(pocs))
    pw, ph = 256, 160
    n_pics = len(pocs)
    _, base = make_pics(rng, bd, pw, ph, BL, (0, 0))
    # every reference = the base displaced in proportion to its POC distance + its
    # own grain; the original = two of them blended, so uni- and bi-directional
    # choices, and different pictures, all win somewhere
    planes = []
    for p in pocs:
        d = p - cur
        shifted = np.roll(base, (d // 3, -d // 2), (0, 1)).astype(np.int32)
        planes.append(np.clip(shifted + rng.integers(-3, 4, base.shape), 0,
                              (1 << bd) - 1).astype(np.uint16))
    mix = (planes[l0[0]].astype(np.int32) + planes[l1[-1]].astype(np.int32) + 1) // 2
    orig = np.clip(np.where(rng.integers(0, 3, (1, base.shape[1])) == 0, planes[l0[-1]], mix) +
                   rng.integers(-2, 3, base.shape), 0, (1 << bd) - 1).astype(np.uint16)
    O = ctx.picture(pw, ph, bd)
    O.upload([orig, None, None], BL)
    R = [ctx.picture(pw, ph, bd) for _ in range(n_pics)]
    for pic, pl in zip(R, planes):
        pic.upload([pl, None, None], BL)
    num_ref = np.array([len(l0), len(l1)], np.int32)
    ref_pic = np.full((2, 3), -1, np.int32)
    ref_pic[0, :len(l0)], ref_pic[1, :len(l1)] = l0, l1
    plane_ptrs = (C.c_void_p * n_pics)(*[pl[BL:, BL:].ctypes.data for pl in planes])
    strides = np.array([pl.strides[0] // 2 for pl in planes], np.int64)
    pocs_a = np.array(pocs, np.int32)
    n = 36
    blocks = np.zeros((2, 3, n), api.ME_DTYPE)
    mvp = np.zeros((2, 3, n, 4), np.int32)
    exp = np.zeros((n, 80), np.int64)
    o = orig[BL:, BL:]
    for i in range(n):
        w, h = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 32, 64]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        fp = int(rng.integers(0, 6) == 0)
        lam = int(rng.choice([120000, 498000, 1500000]))
        nb = np.zeros((2, 2, 3), np.int32)
        nb[:, :, 1:] = rng.integers(-150, 150, (2, 2, 2))
        nb[:, 0, 0] = rng.integers(-1, len(l0), 2)
        nb[:, 1, 0] = rng.integers(-1, len(l1), 2)
        if i % 5 == 0:
            nb[1] = nb[0]
        xr.xr_search_motion_multi(bd, x, y, w, h, fp, lam, iterations, pw, ph, o.ctypes.data,
                                  orig.strides[0] // 2, n_pics, plane_ptrs, strides.ctypes.data,
                                  pocs_a.ctypes.data, cur, num_ref.ctypes.data,
                                  ref_pic.ctypes.data, nb.ctypes.data, exp[i].ctypes.data)
        for l in range(2):
            for r in range(int(num_ref[l])):
                q = exp[i][16 + 8 * (3 * l + r):]
                bk = blocks[l][r][i]
                bk["x"], bk["y"], bk["w"], bk["h"] = x, y, w, h
                bk["depth_nonzero"], bk["fullpel_mv"], bk["lambda16"] = 1, fp, lam
                bk["search_range"] = q[0]
                mvp[l][r][i] = q[1:5]
    same = np.array([exp[0][16 + 8 * (3 + r) + 5] if r < len(l1) else -1 for r in range(3)],
                    np.int32)
    force = int(exp[0][10])
    assert force == (config == "back_only")
    assert [int(v) for v in same[:len(l1)]] == [l0.index(k) if k in l0 else -1 for k in l1]
    handles = (C.c_void_p * 6)(*[R[ref_pic[l][r]].h_pic if ref_pic[l][r] >= 0 else None
                                 for l in range(2) for r in range(3)])
    out = np.zeros((n, 32), np.int64)
    bl, mv = np.ascontiguousarray(blocks), np.ascontiguousarray(mvp)
    assert L.xvc_host_search_motion_multi_batch(
        ctx.h, O.h_pic, handles, num_ref.ctypes.data, same.ctypes.data, 0, force, bl.ctypes.data,
        n, mv.ctypes.data, iterations, out.ctypes.data) == 0
    dirs, refs_used = set(), set()
    for i in range(n):
        e, g = exp[i], out[i]
        d = int(e[1])
        # the uni-directional halves first (cost, ref_idx, mv, mvp_idx per list)
        for l in range(2):
            u = e[64 + 6 * l:70 + 6 * l]
            got = (int(g[10 + l]), int(g[14 + 4 * l]), int(g[16 + 4 * l]), int(g[17 + 4 * l]),
                   int(g[15 + 4 * l]))
            assert got == tuple(int(v) for v in u[:5]), (i, l, tuple(blocks[l][0][i]), got, u)
        assert int(g[12]) == int(e[75]) & 0xffffffff, (i, g[12], e[75])
        want = [d, int(e[0])]
        have = [int(g[0]), int(g[1])]
        for l in range(2):
            if d == 2 or d == l:
                want += [int(e[2 + 4 * l]), int(e[5 + 4 * l]), int(e[3 + 4 * l]), int(e[4 + 4 * l])]
                have += [int(v) for v in g[2 + 4 * l:6 + 4 * l]]
                refs_used.add((l, int(e[2 + 4 * l])))
        if have != want:
            print("CU", i, "host", " ".join(str(int(v)) for v in g[:24]))
            print("CU", i, "ref ", " ".join(str(int(v)) for v in e[:16]), "|",
                  " ".join(str(int(v)) for v in e[64:76]))
        assert have == want, (i, tuple(blocks[0][0][i]))
        dirs.add(d)
    # (input coverage, for the committed seed: a soak seed may make every CU choose bi)
    if not int(os.environ.get("XVC_SOAK", "0")):
        assert 2 in dirs and len(dirs) >= 2, dirs
        assert len({r for _, r in refs_used}) >= 2, refs_used
    for p in [O] + R:
        p.destroy()
