"""GPU parity: every HIP entry point of libxvcgpu.so (called through the C-ABI)
against the CPU oracle on the same seeded inputs.  Bit-exact: all of this path
is integer / byte work (the few double steps are op-for-op identical).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from helpers import SIZES, make_cus, make_pics, random_partition, rnd_samples

pytestmark = pytest.mark.gpu

BL, BC = 128, 64  # device borders


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


@pytest.fixture(scope="module")
def xo():
    return ol.Lib("xo")


def padded_planes(rng, bd, pw, ph, smooth=False):
    planes = []
    for c in range(3):
        w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
        b = BL if c == 0 else BC
        planes.append(rnd_samples(rng, bd, h + 2 * b, w + 2 * b, smooth))
    return planes


def view(planes, c):
    b = BL if c == 0 else BC
    return planes[c][b:, b:]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_metric_batch(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(1000 + bd)
    pw, ph = 256, 192
    for smooth in (False, True):
        pa = padded_planes(rng, bd, pw, ph, smooth)
        pb = padded_planes(rng, bd, pw, ph, smooth)
        A, B = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
        A.upload(pa, BL)
        B.upload(pb, BL)
        for comp in (0, 1):
            cw, ch = (pw, ph) if comp == 0 else (pw // 2, ph // 2)
            cands = []
            for w in [2] + SIZES:
                for h in [2] + SIZES:
                    if min(w, h) == 2 and max(w, h) > 8:
                        continue
                    for metric in range(8):
                        if metric in (4, 6) and h <= 8:
                            continue
                        if metric == 7 and (w < 4 or h < 4 or comp):
                            continue
                        x = int(rng.integers(0, (cw - w) // 2 + 1)) * 2
                        y = int(rng.integers(0, (ch - h) // 2 + 1)) * 2
                        mvx, mvy = int(rng.integers(-40, 40)), int(rng.integers(-40, 40))
                        qp = int(rng.integers(10, 60))
                        cands.append((x, y, w, h, metric, qp, mvx, mvy))
            cands = np.array(cands, api.CAND_DTYPE)
            for weight in ((1.0,) if comp == 0 else (1.0, 0.7937005259840998)):
                got = ctx.metric_batch(A, B, comp, cands, weight=weight)
                va, vb = view(pa, comp), view(pb, comp)
                b = BL if comp == 0 else BC
                for i, cd in enumerate(cands):
                    x, y, w, h = int(cd["x"]), int(cd["y"]), int(cd["w"]), int(cd["h"])
                    a_blk = va[y:y + h, x:x + w]
                    yy, xx = y + int(cd["mv_y"]) + b, x + int(cd["mv_x"]) + b
                    b_blk = pb[comp][yy:yy + h, xx:xx + w]
                    exp = xo.metric_ss(int(cd["metric"]), bd, a_blk, b_blk,
                                       qp=int(cd["qp"]), weight=weight)
                    assert int(got[i]) == exp, (comp, weight, tuple(cd), int(got[i]), exp)
        A.destroy()
        B.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_batch(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(2000 + bd)
    pw, ph = 256, 192
    pr = padded_planes(rng, bd, pw, ph)
    R, P = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(pr, BL)
    blocks = []
    for _ in range(300):
        w = int(rng.choice(SIZES)); h = int(rng.choice(SIZES))
        x = int(rng.integers(0, (pw - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (ph - h) // 4 + 1)) * 4
        comp = int(rng.integers(0, 3))
        r = rng.random()
        if r < 0.15:
            mx, my = int(rng.integers(-8000, 8000)), int(rng.integers(-8000, 8000))
        elif r < 0.3:
            mx, my = int(rng.integers(-40, 40)) * 16, int(rng.integers(-40, 40)) * 16
        elif r < 0.45:
            mx, my = int(rng.integers(-40, 40)) * 16, int(rng.integers(-600, 600))
        else:
            mx, my = int(rng.integers(-600, 600)), int(rng.integers(-600, 600))
        blocks.append((x, y, w, h, comp, 0, mx, my))
    blocks = np.array(blocks, api.MC_DTYPE)
    # blocks may overlap in the pred picture: run one at a time per overlap-free
    # group is overkill; instead run each block alone in a batch of 1..N chunks
    for start in range(0, len(blocks), 1):
        blk = blocks[start:start + 1]
        ctx.mc_batch(R, P, blk)
        b = blk[0]
        comp = int(b["comp"]); cs = 1 if comp else 0
        got = P.download()[comp]
        x, y, w, h = int(b["x"]), int(b["y"]), int(b["w"]), int(b["h"])
        exp = xo.mc_block(bd, comp, x, y, w, h, int(b["mv_x"]), int(b["mv_y"]), pw, ph,
                          pr[comp], BL if comp == 0 else BC)
        g = got[y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs]
        assert np.array_equal(g, exp), tuple(b)
        if start > 120:
            break
    # one big batch over a non-overlapping grid of 16x16 CUs, all components
    grid = []
    for y in range(0, ph, 16):
        for x in range(0, pw, 16):
            mx, my = int(rng.integers(-300, 300)), int(rng.integers(-300, 300))
            for comp in range(3):
                grid.append((x, y, 16, 16, comp, 0, mx, my))
    grid = np.array(grid, api.MC_DTYPE)
    ctx.mc_batch(R, P, grid)
    got = P.download()
    for b in grid:
        comp = int(b["comp"]); cs = 1 if comp else 0
        x, y = int(b["x"]), int(b["y"])
        exp = xo.mc_block(bd, comp, x, y, 16, 16, int(b["mv_x"]), int(b["mv_y"]), pw, ph,
                          pr[comp], BL if comp == 0 else BC)
        g = got[comp][y >> cs:(y + 16) >> cs, x >> cs:(x + 16) >> cs]
        assert np.array_equal(g, exp), tuple(b)
    R.destroy()
    P.destroy()


def me_blocks(rng, api, pw, ph, n, lic=False):
    blocks = np.zeros(n, api.ME_DTYPE)
    for i in range(n):
        w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([4, 8, 16, 32, 64]))
        b = blocks[i]
        b["w"], b["h"] = w, h
        b["x"] = int(rng.integers(0, (pw - w) // 4 + 1)) * 4
        b["y"] = int(rng.integers(0, (ph - h) // 4 + 1)) * 4
        b["depth_nonzero"] = int(rng.integers(0, 2))
        # XVC_ME_FULLPEL_MV | XVC_ME_USE_LIC (AC-only metrics) for a third of the jobs
        b["fullpel_mv"] = int(rng.integers(0, 5) == 0) | (2 if lic and rng.integers(0, 3) == 0 else 0)
        b["mvp_x"] = int(rng.integers(-200, 200))
        b["mvp_y"] = int(rng.integers(-200, 200))
        b["prev_x"] = int(rng.integers(-20, 20))
        b["prev_y"] = int(rng.integers(-20, 20))
        b["lambda16"] = int(rng.choice([120000, 498000, 1500000]))
        b["search_range"] = int(rng.choice([96, 96, 128, 256]))
    return blocks


def to_me_struct(b):
    s = ol.MeBlock()
    for name in ol.ME_DTYPE.names:
        setattr(s, name, int(b[name]))
    return s


@pytest.mark.parametrize("bd", [8, 10])
def test_me_search(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(3000 + bd)
    pw, ph = 320, 192
    for motion in [(3, -2), (0, 0), (-17, 9), (40, 26)]:
        orig, ref = make_pics(rng, bd, pw, ph, BL, motion)
        O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
        O.upload([orig, None, None], BL)
        R.upload([ref, None, None], BL)
        blocks = me_blocks(rng, api, pw, ph, 40, lic=True)
        all_flags = api.ME_FULLPEL | api.ME_SUBPEL | api.ME_LIC_JOBS
        res = ctx.me_search(O, R, blocks, flags=all_flags)
        n_sub = 0
        for i, b in enumerate(blocks):
            s = to_me_struct(b)
            (fx, fy), cost = xo.tz_search(bd, s, pw, ph, orig, ref, BL)
            assert (int(res[i]["fullpel_x"]), int(res[i]["fullpel_y"])) == (fx, fy), \
                (motion, tuple(b), tuple(res[i]), fx, fy)
            assert int(res[i]["fullpel_cost"]) == cost
            if b["fullpel_mv"] & 1:
                assert (int(res[i]["mv_x"]), int(res[i]["mv_y"])) == (fx * 16, fy * 16)
                continue
            (sx, sy), sd = xo.subpel_search(bd, s, pw, ph, orig, ref, BL, (fx, fy))
            assert (int(res[i]["mv_x"]), int(res[i]["mv_y"])) == (sx, sy), \
                (motion, tuple(b), tuple(res[i]), sx, sy)
            assert int(res[i]["subpel_dist"]) == sd
            n_sub += 1
        assert n_sub > 20
        # the two phases run separately give the same answer
        r1 = ctx.me_search(O, R, blocks, flags=api.ME_FULLPEL | api.ME_LIC_JOBS)
        r2 = ctx.me_search(O, R, blocks, flags=api.ME_SUBPEL | api.ME_LIC_JOBS, results=r1)
        assert np.array_equal(r2, res)
        # XVCGPU_ME_HINT_SQ16 picks another kernel for the 16 class (exact-shape instances
        # for 16x16 / 16x8, the any-size instance under its register cap for the rest):
        # a hint changes no result, whatever the sizes in the batch
        r3 = ctx.me_search(O, R, blocks, flags=all_flags | api.ME_HINT_SQ16)
        assert np.array_equal(r3, res)
        sq = blocks.copy()
        sq["w"] = 16
        sq["h"] = np.where(np.arange(len(sq)) % 3 == 0, 8, 16)
        sq["x"] = np.minimum(sq["x"], pw - 16)
        sq["y"] = np.minimum(sq["y"], ph - 16)
        want = ctx.me_search(O, R, sq, flags=all_flags)
        assert np.array_equal(ctx.me_search(O, R, sq, flags=all_flags | api.ME_HINT_SQ16), want)
        # XVCGPU_ME_ONLY_SQ16: the caller's word that every job has such a shape (no second
        # kernel); a job of another shape is answered XVCGPU_ME_UNSUPPORTED, never left as it was
        plain = (sq["fullpel_mv"] & 2) == 0
        got = ctx.me_search(O, R, sq, flags=api.ME_FULLPEL | api.ME_SUBPEL | api.ME_ONLY_SQ16,
                            max_size=16)
        assert np.array_equal(got[plain], ctx.me_search(O, R, sq, flags=api.ME_FULLPEL | api.ME_SUBPEL,
                                                        max_size=16)[plain])
        odd = sq.copy()
        odd["fullpel_mv"] &= 1
        odd["h"][::4] = 4
        got = ctx.me_search(O, R, odd, flags=api.ME_FULLPEL | api.ME_SUBPEL | api.ME_ONLY_SQ16,
                            max_size=16)
        ref_ = ctx.me_search(O, R, odd, flags=api.ME_FULLPEL | api.ME_SUBPEL, max_size=16)
        is_odd = odd["h"] == 4
        assert (got["fullpel_cost"][is_odd] == 0xffffffff).all() and \
            (got["subpel_dist"][is_odd] == 0xffffffff).all()
        assert np.array_equal(got[~is_odd], ref_[~is_odd])
        O.destroy()
        R.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_me_search_extreme_residuals(gpu, xo, bd):
    """Sub-pel SATD at the edge of the packed 16-bit budget (k_subpel.h): the
    original and the reference are opposite Walsh patterns at full swing, so
    every residual is +-(2^bd - 1) and one Hadamard coefficient of each tile
    takes the whole energy, for every tile shape of the fast path."""
    api, ctx = gpu
    rng = np.random.default_rng(3100 + bd)
    pw, ph = 256, 192
    smax = (1 << bd) - 1
    shapes = [(16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (64, 16), (16, 64),
              (8, 8), (16, 16), (32, 32), (64, 64), (32, 8), (8, 32)]
    yy, xx = np.mgrid[0:ph, 0:pw]
    for trial in range(6):
        kx, ky = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        walsh = (np.bitwise_count((xx & kx).astype(np.uint8)).astype(np.int64) +
                 np.bitwise_count((yy & ky).astype(np.uint8))) & 1
        if trial == 5:      # plain sign pattern drawn at random
            walsh = rng.integers(0, 2, (ph, pw))
        orig_in = (walsh * smax).astype(np.uint16)
        ref_in = ((1 - walsh) * smax).astype(np.uint16)
        orig = np.ascontiguousarray(np.pad(orig_in, BL, mode="edge"))
        ref = np.ascontiguousarray(np.pad(ref_in, BL, mode="edge"))
        O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
        O.upload([orig, None, None], BL)
        R.upload([ref, None, None], BL)
        blocks = np.zeros(len(shapes), api.ME_DTYPE)
        for b, (w, h) in zip(blocks, shapes):
            b["w"], b["h"] = w, h
            b["x"] = int(rng.integers(0, (pw - w) // 16 + 1)) * 16
            b["y"] = int(rng.integers(0, (ph - h) // 16 + 1)) * 16
            b["lambda16"], b["search_range"] = 498000, 96
        res = ctx.me_search(O, R, blocks)
        for i, b in enumerate(blocks):
            s = to_me_struct(b)
            (fx, fy), cost = xo.tz_search(bd, s, pw, ph, orig, ref, BL)
            assert (int(res[i]["fullpel_x"]), int(res[i]["fullpel_y"])) == (fx, fy)
            (sx, sy), sd = xo.subpel_search(bd, s, pw, ph, orig, ref, BL, (fx, fy))
            assert (int(res[i]["mv_x"]), int(res[i]["mv_y"])) == (sx, sy), (trial, tuple(b))
            assert int(res[i]["subpel_dist"]) == sd, (trial, tuple(b))
        O.destroy()
        R.destroy()


def test_unsupported_jobs_are_reported(gpu):
    """Job descriptors live in device memory, so a job a kernel cannot take is
    reported in its result slot (the *_UNSUPPORTED records of xvcgpu.h), never
    left as stale memory; the valid jobs beside it are unaffected."""
    api, ctx = gpu
    rng = np.random.default_rng(77)
    pw, ph, bd = 192, 128, 10
    orig, ref = make_pics(rng, bd, pw, ph, BL, (2, 1))
    O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    O.upload([orig, None, None], BL)
    R.upload([ref, None, None], BL)
    blocks = np.zeros(6, api.ME_DTYPE)
    shapes = [(16, 16), (2, 8), (4, 2), (12, 16), (64, 64), (8, 8)]
    for b, (w, h) in zip(blocks, shapes):
        b["x"], b["y"], b["w"], b["h"] = 64, 32, w, h
        b["lambda16"], b["search_range"] = 90000, 96
    stale = np.zeros(6, api.MERES_DTYPE)
    stale["fullpel_cost"] = 12345
    db, dr = ctx.buffer(blocks), ctx.buffer(stale)
    ctx.me_search_dev(O, R, api.ME_FULLPEL | api.ME_SUBPEL, db.ptr, 6, dr.ptr, 16)
    res = dr.to_array(api.MERES_DTYPE, 6)
    for i, ok in enumerate([True, False, False, False, False, True]):   # 64x64 > max 16
        if ok:
            assert res[i]["fullpel_cost"] not in (12345, 0xffffffff), i
        else:
            assert res[i]["fullpel_cost"] == 0xffffffff and res[i]["subpel_dist"] == 0xffffffff, i
    ctx.me_search_dev(O, R, api.ME_FULLPEL | api.ME_SUBPEL, db.ptr, 6, dr.ptr, 64)
    assert dr.to_array(api.MERES_DTYPE, 6)[4]["fullpel_cost"] != 0xffffffff
    # bi-pred refinement: a 32x32 job in a call sized for 16
    jobs = np.zeros(2, api.BI_DTYPE)
    jobs["blk"] = blocks[[0, 4]]
    jobs["blk"]["w"][1] = jobs["blk"]["h"][1] = 32
    dj, dr2 = ctx.buffer(jobs), ctx.buffer(stale[:2])
    ctx.bipred_search_dev(O, R, R, dj.ptr, 2, dr2.ptr, 16)
    r2 = dr2.to_array(api.MERES_DTYPE, 2)
    assert r2[0]["subpel_dist"] != 0xffffffff and r2[1]["subpel_dist"] == 0xffffffff
    # affine ME: an 8-tall block (CanUseAffine needs > 8)
    ab = np.zeros(2, api.AFFINE_ME_DTYPE)
    ab["x"], ab["y"], ab["w"], ab["lambda16"] = 64, 32, 16, 90000
    ab["h"] = [16, 8]
    da, do = ctx.buffer(ab), ctx.alloc(api.AFFINE_ME_RESULT_DTYPE.itemsize * 2)
    ctx._check(ctx.lib.xvcgpu_affine_me_batch(ctx.h, O.h_pic, R.h_pic, None, da.ptr, 2, do.ptr))
    ra = do.to_array(api.AFFINE_ME_RESULT_DTYPE, 2)
    assert ra[0]["dist"] != 0xffffffff and ra[1]["dist"] == 0xffffffff \
        and ra[1]["iterations"] == 0xffffffff
    # intra SATD: a 32x32 job in a call sized for 16
    ij = np.zeros(2, api.INTRA_DTYPE)
    ij["x"], ij["y"], ij["neighbors"] = 64, 32, 7
    ij["w"], ij["h"] = [16, 32], [16, 32]
    di, dd = ctx.buffer(ij), ctx.alloc(4 * api.INTRA_NUM_MODES * 2)
    ctx._check(ctx.lib.xvcgpu_intra_satd_batch(ctx.h, O.h_pic, R.h_pic, di.ptr, 2, dd.ptr, 16))
    dist = dd.to_array(np.uint32, 2 * api.INTRA_NUM_MODES).reshape(2, -1)
    assert (dist[0] != 0xffffffff).all() and (dist[1] == 0xffffffff).all()
    for b in (db, dr, dj, dr2, da, do, di, dd):
        b.free()
    O.destroy()
    R.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_bipred_search(gpu, xo, bd):
    """M2/T2/T7: one SearchBiIterative step per job vs the oracle."""
    api, ctx = gpu
    rng = np.random.default_rng(3500 + bd)
    pw, ph = 320, 192
    for motion in [(2, 1), (-9, 5), (0, 0)]:
        orig, ref_s = make_pics(rng, bd, pw, ph, BL, motion)
        _, ref_o = make_pics(rng, bd, pw, ph, BL, (-motion[0], -motion[1]))
        O, RO, RS = (ctx.picture(pw, ph, bd) for _ in range(3))
        O.upload([orig, None, None], BL)
        RO.upload([ref_o, None, None], BL)
        RS.upload([ref_s, None, None], BL)
        n = 60
        jobs = np.zeros(n, api.BI_DTYPE)
        jobs["blk"] = me_blocks(rng, api, pw, ph, n)
        for j in jobs:
            j["other_mv_x"] = -motion[0] * 16 + int(rng.integers(-100, 100))
            j["other_mv_y"] = -motion[1] * 16 + int(rng.integers(-100, 100))
            j["boot_mv_x"] = motion[0] * 16 + int(rng.integers(-60, 60))
            j["boot_mv_y"] = motion[1] * 16 + int(rng.integers(-60, 60))
        # a few windows clipped by the picture edge (DetermineMinMaxMv)
        jobs[0]["blk"]["x"], jobs[0]["blk"]["y"] = 0, 0
        jobs[0]["boot_mv_x"], jobs[0]["boot_mv_y"] = -(70 << 4), -(75 << 4)
        jobs[1]["blk"]["x"] = pw - int(jobs[1]["blk"]["w"])
        jobs[1]["boot_mv_x"] = 12 << 4
        res = ctx.bipred_search(O, RO, RS, jobs)
        for i, j in enumerate(jobs):
            s = ol.BiBlock()
            for name in ol.ME_DTYPE.names:
                setattr(s.blk, name, int(j["blk"][name]))
            for name in ("other_mv_x", "other_mv_y", "boot_mv_x", "boot_mv_y"):
                setattr(s, name, int(j[name]))
            mv, dist = xo.bipred_search(bd, s, pw, ph, orig, ref_o, ref_s, BL)
            got = ((int(res[i]["mv_x"]), int(res[i]["mv_y"])), int(res[i]["subpel_dist"]))
            assert got == (mv, dist), (motion, i, j, got, mv, dist)
        for p in (O, RO, RS):
            p.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_metric_batch(gpu, xo, bd):
    """T4 building block: motion-compensate then Compare, every metric."""
    api, ctx = gpu
    rng = np.random.default_rng(3600 + bd)
    pw, ph = 256, 192
    orig, ref = make_pics(rng, bd, pw, ph, BL, (3, -2))
    O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    O.upload([orig, None, None], BL)
    R.upload([ref, None, None], BL)
    n = 240
    cands = np.zeros(n, api.MCM_DTYPE)
    for i, c in enumerate(cands):
        w = int(rng.choice(SIZES)); h = int(rng.choice(SIZES))
        c["w"], c["h"] = w, h
        c["x"] = int(rng.integers(0, (pw - w) // 4 + 1)) * 4
        c["y"] = int(rng.integers(0, (ph - h) // 4 + 1)) * 4
        metric = int(rng.integers(0, 8))
        if metric in (4, 6) and h <= 8:
            metric = 3          # the fast variants are only used for h > 8
        c["metric"] = metric
        c["qp"] = int(rng.integers(20, 45))
        lim = 6000 if i % 8 == 0 else 200
        c["mv_x"], c["mv_y"] = int(rng.integers(-lim, lim)), int(rng.integers(-lim, lim))
        if i % 5 == 0:
            c["mv_x"] &= ~15
        if i % 7 == 0:
            c["mv_y"] &= ~15
    got = ctx.mc_metric_batch(O, R, cands, strength=16)
    for i, c in enumerate(cands):
        exp = xo.mc_metric(bd, int(c["metric"]), int(c["qp"]), 16, int(c["x"]), int(c["y"]),
                           int(c["w"]), int(c["h"]), (int(c["mv_x"]), int(c["mv_y"])),
                           pw, ph, orig, ref, BL)
        assert int(got[i]) == exp, (tuple(c), int(got[i]), exp)
    O.destroy()
    R.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_affine_batch(gpu, xo, bd):
    """I3 (affine half): MotionCompAffine for all three components."""
    from test_oracle_vs_ref import affine_mvs
    api, ctx = gpu
    rng = np.random.default_rng(3800 + bd)
    pw, ph = 256, 192
    pr = padded_planes(rng, bd, pw, ph, smooth=True)
    R, P = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(pr, BL)
    grid = []
    i = 0
    y = 0
    for (w, h) in [(64, 64), (32, 64), (64, 32), (32, 32), (16, 16)]:
        if y + h > ph:
            break
        for x in range(0, pw - w + 1, w):
            mv3 = affine_mvs(rng, w, i)
            i += 1
            for comp in range(3):
                grid.append((x, y, w, h, comp, 0, mv3))
        y += h
    grid = np.array(grid, api.MCAFF_DTYPE)
    ctx.mc_affine_batch(R, P, grid)
    got = P.download()
    n_sub = 0
    for b in grid:
        comp = int(b["comp"]); cs = 1 if comp else 0
        x, y, w, h = int(b["x"]), int(b["y"]), int(b["w"]), int(b["h"])
        mv3 = [tuple(int(v) for v in m) for m in b["mv"]]
        exp = xo.mc_affine_block(bd, comp, x, y, w, h, mv3, pw, ph, pr[comp],
                                 BL if comp == 0 else BC)
        g = got[comp][y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs]
        assert np.array_equal(g, exp), (tuple(b), mv3)
        n_sub += mv3[0] != mv3[1]
    assert n_sub > 30
    R.destroy()
    P.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_bipred_batch(gpu, xo, bd):
    """I2: two 14-bit predictions + AddAvg vs the oracle."""
    api, ctx = gpu
    rng = np.random.default_rng(3700 + bd)
    pw, ph = 256, 192
    p0 = padded_planes(rng, bd, pw, ph)
    p1 = padded_planes(rng, bd, pw, ph, smooth=True)
    R0, R1, P = (ctx.picture(pw, ph, bd) for _ in range(3))
    R0.upload(p0, BL)
    R1.upload(p1, BL)
    grid = []
    sizes = [(64, 64), (32, 64), (64, 16), (16, 16), (8, 8), (8, 32), (32, 8), (16, 4),
             (4, 8)]
    y = 0
    for (w, h) in sizes:
        if y + h > ph:
            break
        for x in range(0, pw - w + 1, w):
            r = rng.random()
            lim = 8000 if r < 0.1 else 400
            mv = [int(rng.integers(-lim, lim)) for _ in range(4)]
            if 0.1 <= r < 0.3:
                mv = [v & ~15 for v in mv]           # both full-pel: copy path
            elif r < 0.45:
                mv[0] &= ~15; mv[3] &= ~15           # one direction only
            for comp in range(3):
                grid.append((x, y, w, h, comp, 0, *mv))
        y += h
    grid = np.array(grid, api.MCBI_DTYPE)
    ctx.mc_bipred_batch(R0, R1, P, grid)
    got = P.download()
    for b in grid:
        comp = int(b["comp"]); cs = 1 if comp else 0
        x, y, w, h = int(b["x"]), int(b["y"]), int(b["w"]), int(b["h"])
        exp = xo.mc_bipred_block(bd, comp, x, y, w, h,
                                 (int(b["mv0_x"]), int(b["mv0_y"])),
                                 (int(b["mv1_x"]), int(b["mv1_y"])), pw, ph,
                                 p0[comp], p1[comp], BL if comp == 0 else BC)
        g = got[comp][y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs]
        assert np.array_equal(g, exp), tuple(b)
    assert len(grid) > 100
    for p in (R0, R1, P):
        p.destroy()


def tx_blocks(rng, api, pw, ph, n, with_types=True, restricted=False):
    blocks = np.zeros(n, api.TX_DTYPE)
    for i in range(n):
        comp = int(rng.integers(0, 3))
        cw, ch = (pw, ph) if comp == 0 else (pw // 2, ph // 2)
        sizes = [4, 8, 16, 32, 64] if comp == 0 else [2, 4, 8, 16, 32]
        w = int(rng.choice(sizes)); h = int(rng.choice(sizes))
        b = blocks[i]
        b["w"], b["h"], b["comp"] = w, h, comp
        b["x"] = int(rng.integers(0, (cw - w) // 2 + 1)) * 2
        b["y"] = int(rng.integers(0, (ch - h) // 2 + 1)) * 2
        if with_types and comp == 0 and w >= 4 and h >= 4 and rng.random() < 0.5:
            b["tx_hor"] = int(rng.choice([3, 5]))
            b["tx_ver"] = int(rng.choice([3, 5]))
        elif with_types and comp == 0 and w >= 4 and h >= 4 and rng.random() < 0.2:
            b["tx_hor"] = int(rng.integers(1, 6))
            b["tx_ver"] = int(rng.integers(1, 6))
        if w == 4 and h == 4 and comp == 0 and b["tx_hor"] == 0 and rng.random() < 0.5:
            b["dst4x4"] = 1
        if restricted:
            # what the binding passes under disable_ext2_transform_high_precision:
            # XVC_TX_DCT2_LOW for kDefault / kDct2 of a side of 4..32
            if b["tx_hor"] in (0, 1) and 4 <= w <= 32:
                b["tx_hor"] = 7
            if b["tx_ver"] in (0, 1) and 4 <= h <= 32:
                b["tx_ver"] = 7
        b["qp"] = int(rng.choice([12, 22, 27, 32, 37, 45]))
        # XVC_TXF_*: intra picture, sign hiding off (1 in 5), coefficient scan
        # order (non-diagonal only occurs below 16x16)
        flags = int(rng.integers(0, 2))
        if rng.random() < 0.2:
            flags |= 2
        if max(w, h) < 16 and rng.random() < 0.5:
            flags |= int(rng.integers(1, 3)) << 2
        b["intra_pic"] = flags
    return blocks


def to_tx_struct(b):
    s = ol.TxBlock()
    for name in ol.TX_DTYPE.names:
        setattr(s, name, int(b[name]))
    return s


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_residual_pipeline(gpu, xo, bd):
    _residual_pipeline(gpu, xo, bd, False)


@pytest.mark.parametrize("bd", [8, 10])
def test_residual_pipeline_restricted_mode(gpu, xo, bd):
    """The 6-bit DCT-2 of restricted mode (XVC_TX_DCT2_LOW per direction; the
    oracle's form is pinned to the reference with the restriction flag set in
    test_oracle_vs_ref.py::test_transforms_restricted_mode)."""
    _residual_pipeline(gpu, xo, bd, True)


def _residual_pipeline(gpu, xo, bd, restricted):
    api, ctx = gpu
    rng = np.random.default_rng(4000 + bd + (500 if restricted else 0))
    pw, ph = 256, 128
    n_dist = 0
    for noise in (2, 12, 200):
        po = padded_planes(rng, bd, pw, ph, smooth=True)
        pp = [np.clip(p.astype(np.int32) + rng.integers(-noise, noise + 1, size=p.shape),
                      0, (1 << bd) - 1).astype(np.uint16) for p in po]
        O, P, Rc = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
        O.upload(po, BL)
        P.upload(pp, BL)
        blocks = tx_blocks(rng, api, pw, ph, 60, restricted=restricted)
        assert not restricted or (blocks["tx_hor"] == 7).sum() > 10
        for i, b in enumerate(blocks):
            blk = blocks[i:i + 1]
            Rc.upload(pp, BL)
            levels, off, nnz = ctx.residual_batch(O, P, Rc, blk)
            comp = int(b["comp"])
            ov, pv = view(po, comp), view(pp, comp)
            cw, ch = (pw, ph) if comp == 0 else (pw // 2, ph // 2)
            exp_rec, exp_coeff, exp_n = xo.residual_pipeline(
                bd, to_tx_struct(b), np.ascontiguousarray(ov[:ch, :cw]),
                np.ascontiguousarray(pv[:ch, :cw]))
            w, h = int(b["w"]), int(b["h"])
            assert int(nnz[0]) == exp_n, (noise, tuple(b), int(nnz[0]), exp_n)
            assert np.array_equal(levels.reshape(h, w), exp_coeff), (noise, tuple(b))
            got = Rc.download()[comp]
            assert np.array_equal(got, exp_rec), (noise, tuple(b))
            # split path: forward only == oracle forward; inverse from levels == rec
            coeffs, off2 = ctx.fwd_transform_batch(O, P, blk)
            x, y = int(b["x"]), int(b["y"])
            resi = (ov[y:y + h, x:x + w].astype(np.int32) -
                    pv[y:y + h, x:x + w].astype(np.int32)).astype(np.int16)
            expc = xo.fwd_transform(bd, np.ascontiguousarray(resi), int(b["tx_hor"]),
                                    int(b["tx_ver"]), int(b["dst4x4"]))
            assert np.array_equal(coeffs.reshape(h, w), expc), (noise, tuple(b))
            Rc.upload(pp, BL)
            ctx.inv_transform_batch(P, Rc, blk, levels, off, nnz)
            assert np.array_equal(Rc.download()[comp], exp_rec), (noise, tuple(b))
            # M6: the residual-domain SSD of the coded block (CompareShort on the
            # original and the reconstructed residual, transform_encoder.cc:69-77).
            # Where nothing was clipped the reconstructed residual is rec - pred.
            Rc.upload(pp, BL)
            dist = ctx.inv_transform_dist_batch(O, P, Rc, blk, levels, off, nnz)
            assert np.array_equal(Rc.download()[comp], exp_rec), (noise, tuple(b))
            rb = exp_rec[y:y + h, x:x + w].astype(np.int64)
            if rb.min() > 0 and rb.max() < (1 << bd) - 1:
                d = ov[y:y + h, x:x + w].astype(np.int64) - rb
                assert int(dist[0]) == int((d * d).sum()) >> (2 * (bd - 8)), (noise, tuple(b))
                n_dist += 1
        for pic in (O, P, Rc):
            pic.destroy()
    assert n_dist > 100


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_transform_skip(gpu, xo, bd):
    """X3: blocks <= 4x4 with cu.GetTransformSkip (tx_hor = XVC_TX_SKIP)."""
    api, ctx = gpu
    rng = np.random.default_rng(4300 + bd)
    pw, ph = 64, 64
    po = padded_planes(rng, bd, pw, ph, smooth=True)
    amp = 40 << (bd - 8)
    pp = [np.clip(p.astype(np.int32) + rng.integers(-amp, amp + 1, size=p.shape), 0,
                  (1 << bd) - 1).astype(np.uint16) for p in po]
    O, P, Rc = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    O.upload(po, BL)
    P.upload(pp, BL)
    n = 0
    for comp in range(3):
        for (w, h) in [(4, 4), (2, 2), (2, 4), (4, 2)]:
            if comp == 0 and min(w, h) < 4:
                continue
            for qp in (12, 27, 40):
                cw = pw if comp == 0 else pw // 2
                x = int(rng.integers(0, (cw - w) // 2 + 1)) * 2
                y = int(rng.integers(0, (cw - h) // 2 + 1)) * 2
                blk = np.array([(x, y, w, h, comp, 6, 6, 0, qp, int(rng.integers(0, 2)))],
                               api.TX_DTYPE)
                Rc.upload(pp, BL)
                levels, off, nnz = ctx.residual_batch(O, P, Rc, blk)
                ov, pv = view(po, comp), view(pp, comp)
                exp_rec, exp_coeff, exp_n = xo.residual_pipeline(
                    bd, to_tx_struct(blk[0]), np.ascontiguousarray(ov[:cw, :cw]),
                    np.ascontiguousarray(pv[:cw, :cw]))
                assert int(nnz[0]) == exp_n
                assert np.array_equal(levels.reshape(h, w), exp_coeff)
                assert np.array_equal(Rc.download()[comp], exp_rec), tuple(blk[0])
                coeffs, _ = ctx.fwd_transform_batch(O, P, blk)
                resi = (ov[y:y + h, x:x + w].astype(np.int32) -
                        pv[y:y + h, x:x + w].astype(np.int32)).astype(np.int16)
                assert np.array_equal(coeffs.reshape(h, w),
                                      xo.fwd_transform_skip(bd, np.ascontiguousarray(resi)))
                Rc.upload(pp, BL)
                ctx.inv_transform_batch(P, Rc, blk, levels, off, nnz)
                assert np.array_equal(Rc.download()[comp], exp_rec)
                n += exp_n != 0
    assert n > 5  # the inverse skip path is exercised too
    for pic in (O, P, Rc):
        pic.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_residual_batch_grid(gpu, xo, bd):
    """A whole picture tiled by non-overlapping blocks in one launch."""
    api, ctx = gpu
    rng = np.random.default_rng(4500 + bd)
    pw, ph = 256, 128
    po = padded_planes(rng, bd, pw, ph, smooth=True)
    pp = [np.clip(p.astype(np.int32) + rng.integers(-15, 16, size=p.shape), 0,
                  (1 << bd) - 1).astype(np.uint16) for p in po]
    O, P, Rc = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    O.upload(po, BL)
    P.upload(pp, BL)
    blocks = []
    for (x, y, w, h) in random_partition(rng, pw, ph, 8):
        qp = int(rng.choice([22, 32, 37]))
        blocks.append((x, y, w, h, 0, 0, 0, 0, qp, 0))
        for c in (1, 2):
            blocks.append((x // 2, y // 2, w // 2, h // 2, c, 0, 0, 0, ol.chroma_qp(qp), 0))
    blocks = np.array(blocks, api.TX_DTYPE)
    levels, off, nnz = ctx.residual_batch(O, P, Rc, blocks)
    got = Rc.download()
    exp = [np.ascontiguousarray(view(pp, c)[:(ph if c == 0 else ph // 2),
                                            :(pw if c == 0 else pw // 2)]).copy()
           for c in range(3)]
    for i, b in enumerate(blocks):
        comp = int(b["comp"])
        cw, ch = (pw, ph) if comp == 0 else (pw // 2, ph // 2)
        rec, coeff, n = xo.residual_pipeline(
            bd, to_tx_struct(b), np.ascontiguousarray(view(po, comp)[:ch, :cw]),
            np.ascontiguousarray(view(pp, comp)[:ch, :cw]))
        x, y, w, h = int(b["x"]), int(b["y"]), int(b["w"]), int(b["h"])
        exp[comp][y:y + h, x:x + w] = rec[y:y + h, x:x + w]
        assert int(nnz[i]) == n
        assert np.array_equal(levels[off[i]:off[i] + w * h].reshape(h, w), coeff)
    for c in range(3):
        assert np.array_equal(got[c], exp[c])
    for pic in (O, P, Rc):
        pic.destroy()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bipred", [0, 1])
@pytest.mark.parametrize("sub", [4, 8])
def test_deblock(gpu, xo, bd, bipred, sub):
    api, ctx = gpu
    rng = np.random.default_rng(5000 + bd + bipred + sub)
    total_changed = 0
    for (pw, ph) in [(64, 64), (136, 72), (320, 200)]:
        for trial in range(3):
            parts = random_partition(rng, pw, ph, 4 if sub == 4 else 8)
            l0, l1 = [8, 0, 16][:2 + trial % 2], [16, 8]
            cus, cmap = make_cus(rng, parts, bipred, l0, l1, pw, ph)
            planes = []
            for c in range(3):
                w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
                b = BL if c == 0 else BC
                base = rng.integers(0, 1 << bd, size=((h + 7) // 8, (w + 7) // 8))
                p = np.kron(base, np.ones((8, 8), np.int64))[:h, :w]
                amp = [2, 6, 30][trial]
                p = np.clip(p // [16, 4, 1][trial] + (1 << (bd - 1)) +
                            rng.integers(-amp, amp + 1, size=(h, w)), 0, (1 << bd) - 1)
                full = np.zeros((h + 2 * b, w + 2 * b), np.uint16)
                full[b:b + h, b:b + w] = p
                planes.append(full)
            beta, tc = [(0, 0), (2, -2), (-4, 4)][trial]
            po = [p.copy() for p in planes]
            xo.deblock(bd, pw, ph, bipred, beta, tc, sub, cus, cmap, po, [BL, BC, BC])
            Rc = ctx.picture(pw, ph, bd)
            Rc.upload(planes, BL)
            ctx.deblock(Rc, cus, cmap, bipred, beta, tc, sub)
            got = Rc.download()
            changed = 0
            for c in range(3):
                b = BL if c == 0 else BC
                w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
                assert np.array_equal(got[c], po[c][b:b + h, b:b + w]), (pw, ph, trial, c)
                changed += int((got[c] != planes[c][b:b + h, b:b + w]).sum())
            total_changed += changed
            Rc.destroy()
    assert total_changed > 0     # the filters did change samples


def test_deblock_chains(gpu, xo):
    """All-4-wide / all-4-tall CUs: every edge chains with its neighbour."""
    api, ctx = gpu
    rng = np.random.default_rng(5500)
    bd, pw, ph = 10, 128, 64
    for vertical_strips in (True, False):
        parts = []
        if vertical_strips:
            for x in range(0, pw, 4):
                for y in range(0, ph, 16):
                    parts.append((x, y, 4, 16))
        else:
            for y in range(0, ph, 4):
                for x in range(0, pw, 16):
                    parts.append((x, y, 16, 4))
        cus, cmap = make_cus(rng, parts, 0, [8, 0], [16], pw, ph)
        cus["intra"] = 1
        planes = []
        for c in range(3):
            w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
            b = BL if c == 0 else BC
            full = np.zeros((h + 2 * b, w + 2 * b), np.uint16)
            full[b:b + h, b:b + w] = 512 + rng.integers(-14, 15, size=(h, w))
            planes.append(full)
        po = [p.copy() for p in planes]
        xo.deblock(bd, pw, ph, 0, 0, 0, 4, cus, cmap, po, [BL, BC, BC])
        Rc = ctx.picture(pw, ph, bd)
        Rc.upload(planes, BL)
        ctx.deblock(Rc, cus, cmap, 0, 0, 0, 4)
        got = Rc.download()
        for c in range(3):
            b = BL if c == 0 else BC
            w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
            assert np.array_equal(got[c], po[c][b:b + h, b:b + w]), (vertical_strips, c)
        assert (got[0] != planes[0][BL:BL + ph, BL:BL + pw]).sum() > 100
        Rc.destroy()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bipred", [0, 1])
def test_deblock_pad_ssd(gpu, xo, bd, bipred):
    """The fused tail (k_tail.h): one launch = DeblockPicture (4-sample subblocks,
    CUs >= 8x8) + PadBorder + the luma ComparePicture parts, against the oracle's
    three separate steps; picture sizes with full, remainder and unvisited
    ComparePicture blocks, all tile positions (corner, rim, interior)."""
    api, ctx = gpu
    rng = np.random.default_rng(5600 + bd + bipred)
    total_changed = 0
    sizes = [(64, 64), (136, 72), (320, 200), (352, 288), (8, 8), (72, 200), (640, 384)]
    if bd == 10 and bipred:
        sizes.append((1920, 1080))      # BASELINE config 1's picture: 510 tiles, 7 remainder rows
    for k, (pw, ph) in enumerate(sizes):
        trial = k % 3
        parts = random_partition(rng, pw, ph, 8)
        cus, cmap = make_cus(rng, parts, bipred, [8, 0, 16][:2 + trial % 2], [16, 8], pw, ph)
        planes = []
        for c in range(3):
            w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
            b = BL if c == 0 else BC
            base = rng.integers(0, 1 << bd, size=((h + 7) // 8, (w + 7) // 8))
            p = np.kron(base, np.ones((8, 8), np.int64))[:h, :w]
            amp = [2, 6, 30][trial]
            p = np.clip(p // [16, 4, 1][trial] + (1 << (bd - 1)) +
                        rng.integers(-amp, amp + 1, size=(h, w)), 0, (1 << bd) - 1)
            # the border of the unfiltered picture holds anything: it must not matter
            full = rng.integers(0, 1 << bd, size=(h + 2 * b, w + 2 * b)).astype(np.uint16)
            full[b:b + h, b:b + w] = p
            planes.append(full)
        orig = padded_planes(rng, bd, pw, ph)
        orig[0][BL:BL + ph, BL:BL + pw] = np.clip(
            planes[0][BL:BL + ph, BL:BL + pw].astype(np.int32) +
            rng.integers(-7, 8, size=(ph, pw)), 0, (1 << bd) - 1)
        beta, tc = [(0, 0), (2, -2), (-4, 4)][trial]
        exp = [p.copy() for p in planes]
        xo.deblock(bd, pw, ph, bipred, beta, tc, 4, cus, cmap, exp, [BL, BC, BC])
        xo.pad_border(pw, ph, exp, [BL, BC, BC])
        S, D, O = (ctx.picture(pw, ph, bd) for _ in range(3))
        S.upload(planes, BL)
        O.upload(orig, BL)
        for sbd in (8, bd):
            D.upload([np.zeros_like(p) for p in planes], BL)
            got_ssd = ctx.deblock_pad_ssd(S, D, O, cus, cmap, bipred, beta, tc, sbd)
            got = D.download(BL)
            for c in range(3):
                assert np.array_equal(got[c], exp[c]), (pw, ph, c)
            exp_ssd = xo.picture_ssd(sbd, np.ascontiguousarray(view(orig, 0)[:ph, :pw]),
                                     np.ascontiguousarray(view(exp, 0)[:ph, :pw]))
            assert got_ssd == exp_ssd, (pw, ph, sbd, got_ssd, exp_ssd)
        # without an original: no SSD, same pictures
        D.upload([np.zeros_like(p) for p in planes], BL)
        assert ctx.deblock_pad_ssd(S, D, None, cus, cmap, bipred, beta, tc) is None
        got = D.download(BL)
        for c in range(3):
            assert np.array_equal(got[c], exp[c]), (pw, ph, c)
        total_changed += int((view(exp, 0)[:ph, :pw] != view(planes, 0)[:ph, :pw]).sum())
        assert ctx.lib.xvcgpu_deblock_pad_ssd(ctx.h, S.h_pic, S.h_pic, None, None, 0, None, 0, 0,
                                              0, 0, 8, None) == 10
        for p in (S, D, O):
            p.destroy()
    assert total_changed > 0


def test_pad_border(gpu, xo):
    api, ctx = gpu
    rng = np.random.default_rng(6000)
    for (w, h) in [(64, 64), (136, 72), (352, 288)]:
        planes = padded_planes(rng, 10, w, h)
        P = ctx.picture(w, h, 10)
        P.upload(planes, BL)
        ctx.pad_border(P)
        got = P.download(BL)
        exp = [p.copy() for p in planes]
        xo.pad_border(w, h, exp, [BL, BC, BC])
        for c in range(3):
            assert np.array_equal(got[c], exp[c]), (w, h, c)
        P.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_picture_ssd(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(7000)
    for (w, h) in [(64, 64), (72, 40), (128, 128), (136, 72), (200, 136), (352, 288)]:
        pa = padded_planes(rng, bd, w, h)
        pb = [np.clip(p.astype(np.int32) + rng.integers(-9, 10, size=p.shape), 0,
                      (1 << bd) - 1).astype(np.uint16) for p in pa]
        A, B = ctx.picture(w, h, bd), ctx.picture(w, h, bd)
        A.upload(pa, BL)
        B.upload(pb, BL)
        for comp in (0, 1, 2):
            cw, ch = (w, h) if comp == 0 else (w // 2, h // 2)
            for sbd in (8, bd):
                got = ctx.picture_ssd(A, B, comp, sbd)
                exp = xo.picture_ssd(sbd, np.ascontiguousarray(view(pa, comp)[:ch, :cw]),
                                     np.ascontiguousarray(view(pb, comp)[:ch, :cw]))
                assert got == exp, (w, h, comp, sbd, got, exp)
        A.destroy()
        B.destroy()


def test_error_paths(gpu):
    api, ctx = gpu
    lib = ctx.lib
    p = C.c_void_p()
    assert lib.xvcgpu_picture_create(ctx.h, 60, 64, 10, C.byref(p)) == 10
    assert lib.xvcgpu_picture_create(ctx.h, 64, 64, 7, C.byref(p)) == 10
    assert lib.xvcgpu_picture_create(None, 64, 64, 10, C.byref(p)) == 10
    assert lib.xvcgpu_create(0, None) == 10
    assert lib.xvcgpu_create(99, C.byref(p)) == 20
    A = ctx.picture(64, 64, 10)
    assert lib.xvcgpu_me_search(ctx.h, A.h_pic, A.h_pic, 0, None, 0, None) == 10
    assert lib.xvcgpu_me_search(ctx.h, A.h_pic, A.h_pic, 3, None, 0, None) == 0
    assert lib.xvcgpu_deblock(ctx.h, A.h_pic, None, 0, None, 16, 0, 0, 0, 4) == 10
    # empty batches are accepted, null arrays with n > 0 and mismatching
    # pictures are refused - for every batched entry point
    B = ctx.picture(128, 64, 10)
    d = ctx.alloc(256)
    for name, pics, tail in [
            ("xvcgpu_mc_batch", (A, A), ()),
            ("xvcgpu_mc_affine_batch", (A, A), ()),
            ("xvcgpu_mc_bipred_batch", (A, A, A), ()),
            ("xvcgpu_mc_from_me", (A, A), None)]:
        f = getattr(lib, name)
        if tail is None:
            assert f(ctx.h, A.h_pic, A.h_pic, None, None, 0) == 0
            assert f(ctx.h, A.h_pic, A.h_pic, None, None, 4) == 10
            assert f(ctx.h, A.h_pic, B.h_pic, d.ptr, d.ptr, 1) == 10
            continue
        hs = [p.h_pic for p in pics]
        assert f(ctx.h, *hs, None, 0) == 0, name
        assert f(ctx.h, *hs, None, 3) == 10, name
        assert f(ctx.h, *(hs[:-1] + [B.h_pic]), d.ptr, 1) == 10, name
    assert lib.xvcgpu_bipred_search(ctx.h, A.h_pic, A.h_pic, A.h_pic, None, 0, None, 64) == 0
    assert lib.xvcgpu_bipred_search(ctx.h, A.h_pic, A.h_pic, A.h_pic, None, 2, None, 64) == 10
    assert lib.xvcgpu_bipred_search(ctx.h, A.h_pic, B.h_pic, A.h_pic, d.ptr, 1, d.ptr, 64) == 10
    assert lib.xvcgpu_bipred_search(ctx.h, A.h_pic, A.h_pic, A.h_pic, d.ptr, 1, d.ptr, 3) == 10
    assert lib.xvcgpu_mc_metric_batch(ctx.h, A.h_pic, A.h_pic, 16, None, 0, None) == 0
    assert lib.xvcgpu_mc_metric_batch(ctx.h, A.h_pic, A.h_pic, 16, None, 1, None) == 10
    assert lib.xvcgpu_mc_metric_batch(ctx.h, A.h_pic, B.h_pic, 16, d.ptr, 1, d.ptr) == 10
    assert lib.xvcgpu_residual_batch(ctx.h, A.h_pic, A.h_pic, A.h_pic, None, 0, None, None,
                                     None) == 0
    assert lib.xvcgpu_picture_ssd(ctx.h, A.h_pic, B.h_pic, 0, 10, d.ptr) == 10
    assert lib.xvcgpu_wait_for(ctx.h, None) == 10
    assert lib.xvcgpu_replay(ctx.h, None) == 10
    assert b"mismatch" in lib.xvcgpu_last_error(ctx.h)
    d.free()
    B.destroy()
    A.destroy()


def pad_planes(planes, bd):
    """numpy edge-pad [Y,U,V] to the device border."""
    return [np.ascontiguousarray(np.pad(p, BL if c == 0 else BC, mode="edge"))
            for c, p in enumerate(planes)]


# BASELINE.json configs: CIF QP32, 1080p QP32, 2160p QP27 (+ a ragged size)
def test_frame_pass_rdoq_fused_kernel(gpu, xo):
    """The one-launch form (RDOQ inside recon_from_me: luma wave + U/V half-waves)
    stays available and exact."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, bd, qp = 352, 288, 10, 32
    clip = synth.SyntheticClip(pw, ph, bd)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, fused=True, rdoq=True, rdoq_packed=False)
    assert fp.fused
    ref_host, orig_host = pad_planes(clip.frame(0), bd), pad_planes(clip.frame(1), bd)
    O, R, Rec = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(ref_host, BL)
    O.upload(orig_host, BL)
    fp.run(O, R, Rec)
    ctx.sync()
    res, nnz, cus, ssd = fp.results()
    e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(fp.desc, bd, orig_host, ref_host,
                                                               BL, lib=xo)
    assert np.array_equal(nnz, e_nnz) and np.array_equal(cus, e_cus)
    got = Rec.download(BL)
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    fp.destroy()
    for p in (O, R, Rec):
        p.destroy()


@pytest.mark.parametrize("rdoq", [True, False])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("size", [(352, 288, 32), (136, 72, 32), (1920, 1080, 32),
                                  (3840, 2160, 27)])
def test_frame_pass(gpu, xo, size, fused, rdoq):
    """Whole frame pass (ME -> MC -> residual -> deblock -> pad -> SSD) on the
    GPU against the oracle's frame pass, two chained frames; with the
    reference encoder's quantiser (RDOQ) and with QuantFast."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, qp = size
    if pw > 2000 and not fused:
        pytest.skip("the 2160p case runs the production (fused) path only")
    bd = 10
    clip = synth.SyntheticClip(pw, ph, bd)
    # fused + rdoq: the packed RDOQ kernel between the two halves of the pipeline
    # (the production path); unfused + rdoq: the quantiser inside the residual kernel
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, fused=fused, rdoq=rdoq)
    ref_host = pad_planes(clip.frame(0), bd)
    O, R, Rec = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(ref_host, BL)
    for n in (1, 2):
        orig_host = pad_planes(clip.frame(n), bd)
        O.upload(orig_host, BL)
        fp.run(O, R, Rec, ref_poc=n - 1)
        ctx.sync()
        res, nnz, cus, ssd = fp.results()
        e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(
            fp.desc, bd, orig_host, ref_host, BL, ref_poc=n - 1, lib=xo)
        assert np.array_equal(res, e_res), n
        assert np.array_equal(nnz, e_nnz), n
        assert np.array_equal(cus, e_cus), n
        got = Rec.download(BL)
        for c in range(3):
            assert np.array_equal(got[c], e_rec[c]), (n, c)
        assert (int(ssd[0]), int(ssd[1])) == e_ssd
        assert 25.0 < pipeline.psnr_from_ssd(*e_ssd) < 60.0
        # motion was found: the global pan is (2,1) px/frame
        assert np.median(res["mv_x"]) != 0 or np.median(res["mv_y"]) != 0
        # next frame references this reconstruction
        ref_host = e_rec
        R, Rec = Rec, R
    fp.destroy()
    for p in (O, R, Rec):
        p.destroy()


@pytest.mark.parametrize("size", [(136, 72, 32), (1920, 1080, 32)])
def test_frame_pass_region_major_cu_order(gpu, xo, size):
    """The CU list region by region of a 4 x 2 tiling (pipeline.cu_partition xcd_tiles:
    what bench.py runs, so that an XCD's share of every job list is one compact region):
    the same picture as with the raster list - the order of the jobs changes where a CU's
    results sit, never what they are - and equal to the oracle run on that list."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, qp = size
    bd = 10
    clip = synth.SyntheticClip(pw, ph, bd)
    ref_host, orig_host = pad_planes(clip.frame(0), bd), pad_planes(clip.frame(1), bd)
    O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(ref_host, BL)
    O.upload(orig_host, BL)
    out = {}
    for tiles in (False, True):
        fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, rdoq=True, xcd_tiles=tiles)
        Rec = ctx.picture(pw, ph, bd)
        fp.run(O, R, Rec)
        ctx.sync()
        res, nnz, cus, ssd = fp.results()
        out[tiles] = (Rec.download(BL), (int(ssd[0]), int(ssd[1])), fp.desc.me.copy(), res)
        if tiles:
            e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(
                fp.desc, bd, orig_host, ref_host, BL, lib=xo)
            assert np.array_equal(res, e_res) and np.array_equal(nnz, e_nnz)
            assert np.array_equal(cus, e_cus) and out[True][1] == e_ssd
            for c in range(3):
                assert np.array_equal(out[True][0][c], e_rec[c]), c
        fp.destroy()
        Rec.destroy()
    assert out[False][1] == out[True][1]
    for c in range(3):
        assert np.array_equal(out[False][0][c], out[True][0][c]), c
    # a permutation of the same CUs, each with the same vectors
    key = lambda m: (m["y"].astype(np.int64) << 16) | m["x"]
    a, b = np.argsort(key(out[False][2])), np.argsort(key(out[True][2]))
    assert np.array_equal(out[False][2][a], out[True][2][b])
    assert np.array_equal(out[False][3][a], out[True][3][b])
    if pw > 1000:
        assert not np.array_equal(out[False][2], out[True][2])
    for p in (O, R):
        p.destroy()


@pytest.mark.parametrize("size", [(352, 288, 32), (1920, 1080, 32), (1920, 1080, 40),
                                  (3840, 2160, 27)])
@pytest.mark.parametrize("mode", [0, 1])
def test_frame_pass_all_zero_proof(gpu, xo, size, mode):
    """The same frame passes with the quantiser's all-zero proof
    (xvcgpu_quant_rdo_set_prove_zero) forced on and forced off - by default the
    batch size decides: on at 2160p, off below."""
    api, ctx = gpu
    ctx.set_rdoq_prove_zero(mode)
    try:
        test_frame_pass(gpu, xo, size, True, True)
    finally:
        ctx.set_rdoq_prove_zero(-1)


@pytest.mark.parametrize("rdoq", [False, True])
@pytest.mark.parametrize("n", [2, 3, 4])
def test_frame_pass_multi(gpu, rdoq, n):
    """xvcgpu_frame_pass_multi: n pictures per launch of every kernel = n single
    frame passes (which are pinned against the oracle above): reconstructions
    with their borders, search results, non-zero counts, CU records, SSD - two
    chained rounds, different pictures in every slot."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    pw, ph, qp, bd = 352, 288, 32, 10
    clip = synth.SyntheticClip(pw, ph, bd)
    ctxs = [api.Context(0) for _ in range(n)]
    passes = [pipeline.FramePass(c, pw, ph, bd, qp=qp, rdoq=rdoq) for c in ctxs]
    pipeline.share_stream(passes)
    single = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, rdoq=rdoq)
    pics = [[c.picture(pw, ph, bd) for _ in range(3)] for c in ctxs]      # orig, ref, rec
    sO, sR, sRec = (ctx.picture(pw, ph, bd) for _ in range(3))
    for i in range(n):
        pics[i][1].upload(pad_planes(clip.frame(i), bd), BL)
    for rnd in range(2):
        for i in range(n):
            pics[i][0].upload(pad_planes(clip.frame(i + 1 + rnd), bd), BL)
        pipeline.run_multi(passes, [p[0] for p in pics], [p[1] for p in pics],
                           [p[2] for p in pics], [rnd + i for i in range(n)])
        ctxs[0].sync()
        for i in range(n):
            sO.upload(pics[i][0].download(BL), BL)
            sR.upload(pics[i][1].download(BL), BL)
            single.run(sO, sR, sRec, ref_poc=rnd + i)
            ctx.sync()
            got, exp = pics[i][2].download(BL), sRec.download(BL)
            for c in range(3):
                assert np.array_equal(got[c], exp[c]), (rnd, i, c)
            for a, b in zip(passes[i].results(), single.results()):
                assert np.array_equal(a, b), (rnd, i)
        for p in pics:
            p[1], p[2] = p[2], p[1]          # next round references this reconstruction
    # a form the batched launches do not cover falls back to the single calls
    lone = [passes[0]]
    pipeline.run_multi(lone, [pics[0][0]], [pics[0][1]], [pics[0][2]])
    ctxs[0].sync()
    for p in passes + [single]:
        p.destroy()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("bd,qp,size", [(8, 22, (352, 288)), (8, 37, (200, 120)),
                                        (12, 32, (352, 288)), (12, 17, (136, 72))])
def test_frame_pass_bitdepths(gpu, xo, bd, qp, size):
    """The frame pass at internal bit depths 8 and 12 (the packed sub-pel
    sweep takes bd <= 10; 12 runs the row-major path) and other QPs."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph = size
    clip = synth.SyntheticClip(pw, ph, bd)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp)
    ref_host = pad_planes(clip.frame(0), bd)
    O, R, Rec = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(ref_host, BL)
    for n in (1, 2, 3):
        orig_host = pad_planes(clip.frame(n), bd)
        O.upload(orig_host, BL)
        fp.run(O, R, Rec, ref_poc=n - 1)
        res, nnz, cus, ssd = fp.results()
        e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(
            fp.desc, bd, orig_host, ref_host, BL, ref_poc=n - 1, lib=xo, threads=4)
        assert np.array_equal(res, e_res) and np.array_equal(nnz, e_nnz), n
        assert np.array_equal(cus, e_cus), n
        got = Rec.download(BL)
        for c in range(3):
            assert np.array_equal(got[c], e_rec[c]), (n, c)
        assert (int(ssd[0]), int(ssd[1])) == e_ssd
        ref_host = e_rec
        R, Rec = Rec, R
    fp.destroy()
    for p in (O, R, Rec):
        p.destroy()


@pytest.mark.parametrize("size", [(352, 288), (1920, 1080)])
def test_decode_pass_equals_encoder_reconstruction(gpu, xo, size):
    """N1: the decoder-side reconstruction (MVs + levels -> MC, dequant,
    inverse transform, deblock, pad) reproduces the encoder's reconstruction
    and the oracle's - the reference's own enc-rec == dec-out invariant."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph = size
    bd = 10
    clip = synth.SyntheticClip(pw, ph, bd)
    enc = pipeline.FramePass(ctx, pw, ph, bd, qp=32, keep_levels=True)
    dec = pipeline.DecodePass(ctx, enc.desc, bd)
    ref_host = pad_planes(clip.frame(0), bd)
    O, R, Renc, Rdec = (ctx.picture(pw, ph, bd) for _ in range(4))
    R.upload(ref_host, BL)
    for n in (1, 2):
        orig_host = pad_planes(clip.frame(n), bd)
        O.upload(orig_host, BL)
        enc.run(O, R, Renc, ref_poc=n - 1)
        dec.run(R, Rdec, enc.d_res.ptr, enc.d_levels.ptr, enc.d_level_off.ptr,
                enc.d_nnz.ptr, ref_poc=n - 1)
        ctx.sync()
        a, b = Renc.download(BL), Rdec.download(BL)
        e_rec = oracle_frame.frame_pass(enc.desc, bd, orig_host, ref_host, BL,
                                        ref_poc=n - 1, lib=xo)[0]
        for c in range(3):
            assert np.array_equal(a[c], b[c]), (n, c)
            assert np.array_equal(b[c], e_rec[c]), (n, c)
        levels = enc.d_levels.to_array(np.int16, enc.n_levels)
        assert np.any(levels != 0)
        ref_host = e_rec
        R.upload(ref_host, BL)
    enc.destroy()
    dec.destroy()
    for p in (O, R, Renc, Rdec):
        p.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_bipicture_reconstruction(gpu, xo, bd):
    """N1 for a B picture: CUs predicted from list 0, list 1 or both (random
    partition into 8..64 CUs), residual coded with the device quantiser, then
    decoded from MVs + levels: MC (uni / bi) -> dequant + inverse transform +
    AddClip -> deblocking with two reference POCs per CU -> PadBorder.  The
    encoder-side and decoder-side reconstructions and an oracle composite of
    the pinned block functions must be the same bytes."""
    api, ctx = gpu
    rng = np.random.default_rng(6100 + bd)
    pw, ph, qp = 256, 192, 30
    po = padded_planes(rng, bd, pw, ph, smooth=True)
    r0 = [np.clip(p.astype(np.int32) + rng.integers(-6, 7, size=p.shape), 0,
                  (1 << bd) - 1).astype(np.uint16) for p in po]
    r1 = [np.clip(p.astype(np.int32) + rng.integers(-6, 7, size=p.shape), 0,
                  (1 << bd) - 1).astype(np.uint16) for p in po]
    O, R0, R1, P, Renc, Rdec = (ctx.picture(pw, ph, bd) for _ in range(6))
    O.upload(po, BL)
    R0.upload(r0, BL)
    R1.upload(r1, BL)
    parts = random_partition(rng, pw, ph, 8)
    n = len(parts)
    cus = np.zeros(n, api.CU_DTYPE)
    cmap = -np.ones((ph // 4, pw // 4), np.int32)
    uni0, uni1, bi, tx = [], [], [], []
    qpc = ol.chroma_qp(qp)
    for i, (x, y, w, h) in enumerate(parts):
        d = int(rng.integers(0, 3))               # 0: L0, 1: L1, 2: bi
        mv0 = (int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))
        mv1 = (int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))
        c = cus[i]
        c["x"], c["y"], c["w"], c["h"] = x, y, w, h
        c["qp_y"], c["qp_c"] = qp, qpc
        c["ref_poc"][0] = 0 if d != 1 else -1
        c["ref_poc"][1] = 16 if d != 0 else -1
        if d != 1:
            c["mv"][0][:] = mv0
        if d != 0:
            c["mv"][1][:] = mv1
        cmap[y // 4:(y + h) // 4, x // 4:(x + w) // 4] = i
        for comp in range(3):
            if d == 0:
                uni0.append((x, y, w, h, comp, 0, *mv0))
            elif d == 1:
                uni1.append((x, y, w, h, comp, 0, *mv1))
            else:
                bi.append((x, y, w, h, comp, 0, *mv0, *mv1))
            cs = 1 if comp else 0
            tx.append((x >> cs, y >> cs, w >> cs, h >> cs, comp, 0, 0, 0,
                       qpc if comp else qp, 0))
    assert uni0 and uni1 and bi
    tx = np.array(tx, api.TX_DTYPE)

    def predict(dst):
        ctx.mc_batch(R0, dst, np.array(uni0, api.MC_DTYPE))
        ctx.mc_batch(R1, dst, np.array(uni1, api.MC_DTYPE))
        ctx.mc_bipred_batch(R0, R1, dst, np.array(bi, api.MCBI_DTYPE))

    # encoder side
    predict(P)
    levels, off, nnz = ctx.residual_batch(O, P, Renc, tx)
    cus["cbf_luma"] = nnz[0::3] != 0
    ctx.deblock(Renc, cus, cmap, bipred=1)
    ctx.pad_border(Renc)
    ctx.sync()
    # decoder side: from MVs + levels only
    predict(P)
    ctx.inv_transform_batch(P, Rdec, tx, levels, off, nnz)
    ctx.deblock(Rdec, cus, cmap, bipred=1)
    ctx.pad_border(Rdec)
    ctx.sync()
    enc, dec = Renc.download(BL), Rdec.download(BL)
    # oracle composite
    pred = [np.zeros_like(p) for p in po]
    for (x, y, w, h, comp, _, mx, my) in uni0:
        b = BL if comp == 0 else BC
        cs = 1 if comp else 0
        view(pred, comp)[y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs] = \
            xo.mc_block(bd, comp, x, y, w, h, mx, my, pw, ph, r0[comp], b)
    for (x, y, w, h, comp, _, mx, my) in uni1:
        b = BL if comp == 0 else BC
        cs = 1 if comp else 0
        view(pred, comp)[y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs] = \
            xo.mc_block(bd, comp, x, y, w, h, mx, my, pw, ph, r1[comp], b)
    for (x, y, w, h, comp, _, a0, a1, b0, b1) in bi:
        b = BL if comp == 0 else BC
        cs = 1 if comp else 0
        view(pred, comp)[y >> cs:(y + h) >> cs, x >> cs:(x + w) >> cs] = \
            xo.mc_bipred_block(bd, comp, x, y, w, h, (a0, a1), (b0, b1), pw, ph,
                               r0[comp], r1[comp], b)
    rec = [p.copy() for p in pred]
    for k, t in enumerate(tx):
        comp = int(t["comp"])
        cw, ch = (pw, ph) if comp == 0 else (pw // 2, ph // 2)
        r, coeff, nz = xo.residual_pipeline(
            bd, to_tx_struct(t), np.ascontiguousarray(view(po, comp)[:ch, :cw]),
            np.ascontiguousarray(view(pred, comp)[:ch, :cw]))
        assert nz == int(nnz[k])
        x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
        view(rec, comp)[y:y + h, x:x + w] = r[y:y + h, x:x + w]
    xo.deblock(bd, pw, ph, 1, 0, 0, 4, cus, cmap, rec, [BL, BC, BC])
    xo.pad_border(pw, ph, rec, [BL, BC, BC])
    for c in range(3):
        assert np.array_equal(enc[c], dec[c]), c
        assert np.array_equal(dec[c], rec[c]), c
    for p in (O, R0, R1, P, Renc, Rdec):
        p.destroy()


def test_recorded_frame_pass_replay(gpu, xo):
    """xvcgpu_record_* / xvcgpu_replay: the recorded frame pass replayed as one
    HIP graph gives the same bytes as the oracle, on every replay."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, bd = 208, 112, 10
    clip = synth.SyntheticClip(pw, ph, bd)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=32)
    ref_host = pad_planes(clip.frame(0), bd)
    orig_host = pad_planes(clip.frame(1), bd)
    O, R, Rec = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    R.upload(ref_host, BL)
    O.upload(orig_host, BL)
    ctx.sync()
    rec = ctx.record(lambda: fp.run(O, R, Rec, ref_poc=0))
    # nothing ran while recording
    assert not np.any(Rec.download()[0])
    e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(
        fp.desc, bd, orig_host, ref_host, BL, ref_poc=0, lib=xo)
    for _ in range(3):
        ctx.replay(rec)
        ctx.sync()
        res, nnz, cus, ssd = fp.results()
        assert np.array_equal(res, e_res) and np.array_equal(cus, e_cus)
        got = Rec.download(BL)
        for c in range(3):
            assert np.array_equal(got[c], e_rec[c]), c
        assert (int(ssd[0]), int(ssd[1])) == e_ssd
    ctx.recording_destroy(rec)
    # synchronising calls are refused while recording
    ctx._check(ctx.lib.xvcgpu_record_begin(ctx.h))
    h = C.c_void_p()
    assert ctx.lib.xvcgpu_record_end(ctx.h, C.byref(h)) == 0
    ctx.recording_destroy(h)
    fp.destroy()
    for p in (O, R, Rec):
        p.destroy()


@pytest.mark.parametrize("size", [(352, 288), (1920, 1080)])
def test_pipelined_frame_pass(gpu, xo, size):
    """Two-queue issue of the frame pass (high / low priority streams, halves
    of the picture): identical bytes, three chained pictures."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph = size
    bd = 10
    lo = api.Context(0)
    ctx.use_priority_stream(True)
    lo.use_priority_stream(False)
    clip = synth.SyntheticClip(pw, ph, bd)
    fp = pipeline.PipelinedFramePass(ctx, lo, pw, ph, bd, qp=32)
    ref_host = pad_planes(clip.frame(0), bd)
    O = [ctx.picture(pw, ph, bd) for _ in range(3)]
    recs = [ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)]
    recs[0].upload(ref_host, BL)
    origs = [pad_planes(clip.frame(n), bd) for n in (1, 2, 3)]
    for n in range(3):
        O[n].upload(origs[n], BL)
    ctx.sync()
    # all three pictures queued back to back: cross-picture ordering is part of the test
    for n in range(3):
        fp.run(O[n], recs[n % 2], recs[(n + 1) % 2], ref_poc=n)
    fp.sync()
    exp_ref = ref_host
    for n in range(3):
        e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(
            fp.desc, bd, origs[n], exp_ref, BL, ref_poc=n, lib=xo)
        exp_ref = e_rec
    got = recs[1].download(BL)        # picture 3 landed in recs[(2 + 1) % 2]
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    res, nnz, cus, ssd = fp.results()
    assert np.array_equal(res, e_res) and np.array_equal(nnz, e_nnz)
    assert np.array_equal(cus, e_cus)
    assert (int(ssd[0]), int(ssd[1])) == e_ssd
    fp.destroy()
    for p in O + recs:
        p.destroy()
    lo.close()
    ctx.use_own_stream()


@pytest.mark.parametrize("cu", [8, 32, 64])
def test_frame_pass_cu_sizes(gpu, xo, cu):
    """Frame pass with other CU sizes (8: fused wave kernels; 32 / 64: the
    32- and 64-class search instances and the workgroup residual kernel)."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, bd = 352, 288, 10
    clip = synth.SyntheticClip(pw, ph, bd)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=30, cu=cu)
    ref_host, orig_host = pad_planes(clip.frame(0), bd), pad_planes(clip.frame(2), bd)
    O, R, Rec = (ctx.picture(pw, ph, bd) for _ in range(3))
    R.upload(ref_host, BL)
    O.upload(orig_host, BL)
    fp.run(O, R, Rec)
    ctx.sync()
    res, nnz, cus, ssd = fp.results()
    e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(fp.desc, bd, orig_host,
                                                                ref_host, BL, lib=xo)
    assert np.array_equal(res, e_res) and np.array_equal(nnz, e_nnz)
    assert np.array_equal(cus, e_cus)
    got = Rec.download(BL)
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    assert (int(ssd[0]), int(ssd[1])) == e_ssd
    fp.destroy()
    for p in (O, R, Rec):
        p.destroy()


def test_frame_pass_8k_10bit_qp37(gpu, xo):
    """BASELINE config 5 (7680x4320 10-bit, QP 37) at full size: one frame pass
    against the oracle, and the decoder-side pass reproduces it."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_frame
    pw, ph, bd, qp = 7680, 4320, 10, 37
    clip = synth.SyntheticClip(pw, ph, bd)
    ref_host, orig_host = pad_planes(clip.frame(0), bd), pad_planes(clip.frame(1), bd)
    O, R, Renc, Rdec = (ctx.picture(pw, ph, bd) for _ in range(4))
    R.upload(ref_host, BL)
    O.upload(orig_host, BL)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp)
    fp.run(O, R, Renc)
    ctx.sync()
    e_rec, e_res, _, _, e_ssd = oracle_frame.frame_pass(fp.desc, bd, orig_host, ref_host,
                                                        BL, lib=xo)
    res, _, _, ssd = fp.results()
    assert np.array_equal(res, e_res)
    got = Renc.download(BL)
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    assert (int(ssd[0]), int(ssd[1])) == e_ssd
    fp.destroy()
    enc = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, keep_levels=True)
    dec = pipeline.DecodePass(ctx, enc.desc, bd)
    enc.run(O, R, Renc)
    dec.run(R, Rdec, enc.d_res.ptr, enc.d_levels.ptr, enc.d_level_off.ptr, enc.d_nnz.ptr)
    ctx.sync()
    a = Rdec.download(BL)
    for c in range(3):
        assert np.array_equal(a[c], e_rec[c]), c
    enc.destroy()
    dec.destroy()
    for p in (O, R, Renc, Rdec):
        p.destroy()


# (208x112, 2/3 shards) small; (3840x2160 QP32, 8 shards) = BASELINE config 4:
# there a shard keeps only the rows its next search can reach (no all-gather)
@pytest.mark.parametrize("world", [2, 3, 8, 108])
def test_sharded_gpu_engine_loopback(gpu, xo, world):
    """The multi-GPU orchestration with the real HIP engine: `world` shards of
    one picture handled by separate GpuEngine instances on this one GPU, data
    exchanged by device-to-device copies instead of RCCL.  Must equal the
    unsharded oracle frame pass bit for bit."""
    import torch
    import oracle_frame
    from test_sharded import LoopbackComm, assert_valid_rows_equal
    from xvc_amd import pipeline, sharded, synth
    api, ctx = gpu
    pw, ph, bd, qp = (3840, 2160, 10, 32) if world == 8 else (208, 112, 10, 32)
    if world == 108:        # 8 shards of 1080p: the driver's 8-GPU run
        world, pw, ph = 8, 1920, 1080
    dev = torch.device("cuda", 0)
    clip = synth.SyntheticClip(pw, ph, bd)
    desc = pipeline.FrameDescriptors(pw, ph, qp)
    rows = sharded.shard_rows(ph, world)
    ranks = []
    for r in range(world):
        e = sharded.GpuEngine(ctx, pw, ph, bd, qp, rows[r], dev)
        e.pictures[0].upload(pad_planes(clip.frame(0), bd), BL)
        ranks.append(sharded.ShardedFramePass(e, LoopbackComm(), r, world))
    O = ctx.picture(pw, ph, bd)
    ref_host = pad_planes(clip.frame(0), bd)
    if world == 8:
        assert ranks[0].valid_rows() != (0, ph)
        sends, recvs = ranks[0].gather_ops(0)
        assert {p for p, _ in sends} | {p for p, _ in recvs} == {1}
    for n in ((1, 2, 3) if ph == 1080 else (1, 2)):
        orig_host = pad_planes(clip.frame(n), bd)
        O.upload(orig_host, BL)
        ref_idx, rec_idx = (n - 1) % 2, n % 2
        for s in ranks:
            s.phase_a(O, ref_idx, rec_idx, n - 1)
        LoopbackComm.exchange_all({s.rank: s.halo_ops(rec_idx) for s in ranks})
        for s in ranks:
            s.phase_b(rec_idx)
        LoopbackComm.exchange_all({s.rank: s.gather_ops(rec_idx) for s in ranks})
        for s in ranks:
            s.phase_c(O, rec_idx)
        torch.cuda.synchronize()
        ctx.sync()
        e_rec, _, _, _, e_ssd = oracle_frame.frame_pass(desc, bd, orig_host, ref_host, BL,
                                                        n - 1, lib=xo)
        total = [0, 0]
        for s in ranks:
            got = s.e.pictures[rec_idx].download(BL)
            assert_valid_rows_equal(s, got, e_rec, ph, (world, n))
            part = s.e.ssd_tensor().cpu()
            total[0] += int(part[0])
            total[1] += int(part[1])
        assert tuple(total) == e_ssd
        ref_host = e_rec
    ctx.use_own_stream()
    O.destroy()


def test_sharded_rccl_single_rank(gpu):
    """GpuEngine + TorchComm on the RCCL backend (process group init, barrier,
    int64 all-reduce of the PSNR parts) as a one-rank job in its own process."""
    import socket
    import subprocess
    import sys
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "nccl_one_rank.py"), str(port)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout.split(), r.stdout + r.stderr


def test_concurrent_chains(gpu, xo):
    """bench.py --chains: independent picture chains issued round-robin on
    their own contexts / streams run concurrently on the device; every chain's
    reconstructions and PSNR sums equal the oracle's for that chain (no
    cross-talk through scratch buffers or scheduling records)."""
    import oracle_frame
    from xvc_amd import pipeline, synth
    api, _ = gpu
    pw, ph, bd, qp, n_chains, n_frames = 352, 288, 10, 32, 3, 4
    clip = synth.SyntheticClip(pw, ph, bd)
    desc = pipeline.FrameDescriptors(pw, ph, qp)
    ctxs = [api.Context(0) for _ in range(n_chains)]
    chains = []
    for c, ctx in enumerate(ctxs):
        frames = [pad_planes(clip.frame(3 * c + n), bd) for n in range(n_frames + 1)]
        origs = []
        for f in frames[1:]:
            p = ctx.picture(pw, ph, bd)
            p.upload(f, BL)
            origs.append(p)
        recs = [ctx.picture(pw, ph, bd) for _ in range(n_frames + 1)]
        recs[0].upload(frames[0], BL)
        chains.append((ctx, pipeline.FramePass(ctx, pw, ph, bd, qp=qp), origs, recs, frames))
    for rep in range(3):       # several rounds so that the launches really interleave
        for n in range(n_frames):
            for ctx, fp, origs, recs, _ in chains:
                fp.run(origs[n], recs[n], recs[n + 1], ref_poc=n)
    for ctx in ctxs:
        ctx.sync()
    for ctx, fp, origs, recs, frames in chains:
        ref = frames[0]
        for n in range(n_frames):
            e_rec, _, _, _, e_ssd = oracle_frame.frame_pass(desc, bd, frames[n + 1], ref, BL, n,
                                                            lib=xo, threads=4)
            got = recs[n + 1].download(BL)
            for c in range(3):
                assert np.array_equal(got[c], e_rec[c]), (n, c)
            ref = e_rec
        ssd = fp.d_ssd.to_array(np.uint64, 2)
        assert (int(ssd[0]), int(ssd[1])) == e_ssd
        fp.destroy()
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_lic_batch(gpu, xo, bd):
    """Motion compensation with local illumination compensation (I3, LIC
    half) against the oracle: random CUs / neighbours / vectors, all three
    components, gain and offset changes between reference and current picture."""
    import oracle_lic as ol_
    api, ctx = gpu
    rng = np.random.default_rng(1100 + bd)
    pw, ph = 256, 192
    mx = (1 << bd) - 1
    total = 0
    for content in range(3):
        cur, ref = make_pics(rng, bd, pw, ph, BL, motion=(2, 1), noise=3)
        gain = [1.0, 0.8, 1.3][content]
        rec_y = np.clip(cur[BL:BL + ph, BL:BL + pw].astype(np.float64) * gain + 9 * content,
                        0, mx).astype(np.uint16)
        chroma_ref = [rnd_samples(rng, bd, ph // 2 + 2 * BC, pw // 2 + 2 * BC, True)
                      for _ in range(2)]
        rec_c = [np.clip(c[BC:BC + ph // 2, BC:BC + pw // 2].astype(np.int64) * 7 // 8 + 30, 0,
                         mx).astype(np.uint16) for c in chroma_ref]
        ref_planes = [np.ascontiguousarray(ref)] + chroma_ref
        rec_planes = [np.ascontiguousarray(rec_y)] + [np.ascontiguousarray(c) for c in rec_c]
        R, C_, P = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
        R.upload(ref_planes, BL)
        C_.upload(rec_planes)
        jobs = [j for j, _, _ in ol_.random_jobs(rng, pw, ph, 80)]
        for j in jobs:           # blocks overlap: one launch per job
            ctx.mc_lic_batch(R, C_, P, np.array([j], api.LIC_DTYPE))
            c, s = int(j["comp"]), 1 if j["comp"] else 0
            x, y, w, h = int(j["x"]) >> s, int(j["y"]) >> s, int(j["w"]) >> s, int(j["h"]) >> s
            exp = ol_.xo_mc_lic(xo, bd, j, pw, ph, ref_planes, [BL, BC, BC], rec_planes)
            got = P.download()[c]
            assert np.array_equal(got[y:y + h, x:x + w], exp[y:y + h, x:x + w]), (content, j)
            total += 1
        for p in (R, C_, P):
            p.destroy()
    assert total == 240


def test_timer_slots_and_stream_handle(gpu):
    """xvcgpu_timer_mark / _between (several intervals in flight) and
    xvcgpu_get_stream."""
    api, ctx = gpu
    P = ctx.picture(1920, 1080, 10)
    for rep in range(2):        # the first round creates the events and warms up
        for k in range(4):
            ctx.timer_mark(2 * k)
            for _ in range(4 * k + 1):
                ctx.pad_border(P)
            ctx.timer_mark(2 * k + 1)
        ctx.sync()
    ms = [ctx.timer_between(2 * k, 2 * k + 1) for k in range(4)]
    assert all(m > 0 for m in ms) and ms[3] > ms[0]
    assert ctx.timer_between(0, 7) >= sum(ms) * 0.9
    lib = ctx.lib
    f = C.c_float(0)
    assert lib.xvcgpu_timer_mark(ctx.h, 64) == 10
    assert lib.xvcgpu_timer_between(ctx.h, 0, 40, C.byref(f)) == 10   # slot never recorded
    assert ctx.stream_ptr() != 0
    ctx2 = api.Context(0)
    assert ctx2.stream_ptr() not in (0, ctx.stream_ptr())
    ctx2.close()
    P.destroy()


# (8 ranks at 1080p = the shard plan of the driver's 8-GPU run: 128 / 144-row shards,
# rows exchanged with up to two neighbours on each side)
@pytest.mark.parametrize("world,size", [(2, (352, 288)), (3, (352, 288)), (8, (1920, 1080))])
def test_sharded_ranks_share_one_gpu(gpu, world, size):
    """GpuEngine + TorchComm as separate processes that exchange device
    tensors (gloo transport: RCCL refuses several ranks on one device): halo
    exchange, neighbour-limited row exchange and the all-reduced PSNR parts of
    the row-sharded pass, every rank against the oracle."""
    import socket
    import subprocess
    import sys
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "gloo_two_ranks_gpu.py"),
                               str(r), str(world), str(port), str(size[0]), str(size[1])],
                              stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("OK rank %d" % r) in o, o[-3000:]


def test_bench_multi_rank_path_runs(gpu):
    """bench.py's N > 1 branch end to end (row shards, three picture chains
    with their own process groups and streams, exchanges, all-reduced PSNR,
    max-over-ranks timing) as two ranks on this one GPU over gloo - the
    functional check of what the driver launches on a multi-GPU node."""
    import json
    import socket
    import subprocess
    import sys
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XVC_BENCH_DEVICE="0", XVC_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "24", "--warmup", "6", "--no-cpu",
                        "--width", "352", "--height", "288", "--schedule", "rows"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 24 and d["scaling"] == "strong"
    assert d["pictures_in_flight"] == 3 and 25.0 < d["psnr_y"] < 60.0
    assert "cu-row-shard2" in d["config"]["parallelism"]
    # what makes the line readable on its own: the exchanges' RCCL operations and bytes per
    # picture from the C++ plan, and the CPU baseline quoted from the N = 1 run
    ex = d["shard_exchange_per_picture"]
    assert ex and all(len(v) == 2 and v[1] > 0 for v in ex.values())
    assert d["cpu_baseline"] is None or "quoted_from" in d["cpu_baseline"]


@pytest.mark.parametrize("size,qp", [((352, 288), 30), ((1920, 1080), 36)])
def test_frame_pass_proof_with_random_context_states(gpu, size, qp):
    """The proof inside the forward kernel with context snapshots other than the
    picture-initial ones: a frame pass whose quantiser reads random context states,
    proof off against proof on - levels, counts, CU records and reconstruction equal
    (the walk itself is held against the oracle with random states in test_gpu_rdoq.py)."""
    api, ctx = gpu
    from xvc_amd import pipeline, synth
    import oracle_rdoq as oq
    pw, ph = size
    bd = 10
    clip = synth.SyntheticClip(pw, ph, bd)
    rng = np.random.default_rng(int(os.environ.get("XVC_SOAK", 0)) * 7919 + 99)
    fp = pipeline.FramePass(ctx, pw, ph, bd, qp=qp, rdoq=True)
    O, R = ctx.picture(pw, ph, bd), ctx.picture(pw, ph, bd)
    recs = [ctx.picture(pw, ph, bd) for _ in range(2)]
    R.upload(pad_planes(clip.frame(0), bd), BL)
    O.upload(pad_planes(clip.frame(1), bd), BL)
    proved_total = 0
    for trial in range(3):
        snap = oq.random_contexts(rng)
        ctx._check(ctx.lib.xvcgpu_memcpy_h2d(ctx.h, fp.d_rdoq_ctx.ptr, snap.ctypes.data,
                                             snap.nbytes))
        out = []
        for mode, rec in zip((0, -1), recs):
            ctx.set_rdoq_prove_zero(mode)
            fp.run(O, R, rec, ref_poc=0)
            ctx.sync()
            res, nnz, cus, ssd = fp.results()
            lv = fp.d_levels.to_array(np.int16, fp.n_levels)
            cc = (C.c_int32 * 3)()
            ctx._check(ctx.lib.xvcgpu_quant_rdo_class_counts(ctx.h, cc))
            out.append((nnz, cus, lv, rec.download(BL), tuple(int(v) for v in ssd), sum(cc)))
        ctx.set_rdoq_prove_zero(-1)
        a, b = out
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), trial
        assert np.array_equal(a[2], b[2]), trial
        for c in range(3):
            assert np.array_equal(a[3][c], b[3][c]), (trial, c)
        assert a[4] == b[4]
        proved_total += a[5] - b[5]
    assert proved_total > 0
    fp.destroy()
    for p in [O, R] + recs:
        p.destroy()
