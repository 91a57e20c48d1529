"""Pins the CPU oracle (oracle/xvc_oracle*.c) against the reference's own
compiled code (oracle/_ref/libxvcref.so) on seeded random inputs.

Runs only where the reference harness has been built (this container:
`make -C oracle ref`); elsewhere the committed golden vectors
(test_oracle_golden.py) carry the same pin.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from helpers import SIZES, make_cus, make_pics, random_partition, rnd_samples

pytestmark = pytest.mark.skipif(not ol.have_ref(),
                                reason="reference harness not built")



@pytest.fixture(scope="module")
def libs():
    return ol.Lib("xo"), ol.Lib("xr")


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("simd", [0, 1])
def test_metrics_sample_sample(libs, bd, simd):
    xo, xr = libs
    xr._set_simd(simd)
    rng = np.random.default_rng(100 + bd)
    n = 0
    for w in [2] + SIZES:
        for h in [2] + SIZES:
            if (w == 2) != (h == 2) and min(w, h) == 2 and max(w, h) > 8:
                continue
            for smooth in (False, True):
                a = rnd_samples(rng, bd, h + 3, w + 5, smooth)[1:h + 1, 2:w + 2]
                b = rnd_samples(rng, bd, h + 2, w + 9, smooth)[1:h + 1, 3:w + 3]
                for metric in range(8):
                    if metric in (4, 6) and h <= 8:  # Fast only used for H > 8 (inter_search.cc:1067)
                        continue
                    if metric == 7 and (w < 4 or h < 4):
                        continue
                    for qp in ((32,) if metric != 7 else (12, 27, 32, 45, 63)):
                        r = xr.metric_ss(metric, bd, a, b, qp=qp)
                        o = xo.metric_ss(metric, bd, a, b, qp=qp)
                        assert r == o, (metric, bd, w, h, smooth, qp)
                        n += 1
    xr._set_simd(1)
    assert n > 500


@pytest.mark.parametrize("bd", [8, 10])
def test_metrics_residual_sample(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(7 + bd)
    for w in SIZES:
        for h in SIZES:
            # "2*orig - pred" range
            a = rng.integers(-(1 << bd), 2 << bd, size=(h, 64)).astype(np.int16)[:, :w]
            b = rnd_samples(rng, bd, h, w + 7)[:, 3:w + 3]
            for metric in range(7):
                if metric in (4, 6) and h <= 8:
                    continue
                assert xr.metric_rs(metric, bd, a, b) == xo.metric_rs(metric, bd, a, b), \
                    (metric, w, h)
            a2 = rng.integers(-2000, 2000, size=(h, 64)).astype(np.int16)[:, :w]
            b2 = rng.integers(-2000, 2000, size=(h, 64)).astype(np.int16)[:, :w]
            assert xr.ssd_rr(bd, a2, b2) == xo.ssd_rr(bd, a2, b2)


@pytest.mark.parametrize("bd", [8, 10])
def test_picture_ssd(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(3)
    for (w, h) in [(64, 64), (72, 40), (128, 128), (136, 72), (200, 136), (352, 288)]:
        a = rnd_samples(rng, bd, h, w)
        b = np.clip(a.astype(np.int32) + rng.integers(-9, 10, size=(h, w)), 0,
                    (1 << bd) - 1).astype(np.uint16)
        ro, rn = xr.picture_ssd(bd, a, b)
        oo, on = xo.picture_ssd(bd, a, b)
        assert ro == oo and rn == on, (w, h, ro, oo, rn, on)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("simd", [0, 1])
def test_interpolation(libs, bd, simd):
    xo, xr = libs
    xr._set_simd(simd)
    rng = np.random.default_rng(11 + bd)
    for is_chroma in (0, 1):
        nph = 32 if is_chroma else 16
        for w in ([2] if is_chroma else []) + SIZES:
            for h in ([2] if is_chroma else []) + SIZES[:4]:
                plane = rnd_samples(rng, bd, h + 16, w + 16)
                fracs = [(0, 0), (0, 5), (7, 0)] + \
                    [(int(rng.integers(1, nph)), int(rng.integers(1, nph))) for _ in range(3)]
                for fx, fy in fracs:
                    for bip in (False, True):
                        r = xr.mc_uni(bd, is_chroma, w, h, fx, fy, plane, 8, 8, bip)
                        o = xo.mc_uni(bd, is_chroma, w, h, fx, fy, plane, 8, 8, bip)
                        assert np.array_equal(r, o), (is_chroma, w, h, fx, fy, bip)
    # bi-pred average
    for w in [2] + SIZES:
        a = rng.integers(-8192, 8191, size=(8, 64)).astype(np.int16)[:, :w]
        b = rng.integers(-8192, 8191, size=(8, 64)).astype(np.int16)[:, :w]
        assert np.array_equal(xr.add_avg(bd, a, b), xo.add_avg(bd, a, b))
    xr._set_simd(1)


def test_clip_and_window(libs):
    xo, xr = libs
    rng = np.random.default_rng(5)
    for _ in range(300):
        pw, ph = int(rng.choice([64, 352, 1920])), int(rng.choice([64, 288, 1080]))
        x = int(rng.integers(0, pw // 8)) * 8
        y = int(rng.integers(0, ph // 8)) * 8
        mx, my = int(rng.integers(-40000, 40000)), int(rng.integers(-40000, 40000))
        assert xr.clip_mv(x, y, pw, ph, mx, my) == xo.clip_mv(x, y, pw, ph, mx, my)
        rng_ = int(rng.choice([4, 96, 256]))
        a = xr.min_max_mv(x, y, pw, ph, mx // 8, my // 8, rng_)
        b = xo.min_max_mv(x, y, pw, ph, mx // 8, my // 8, rng_)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert xr._mvd_bits_fullpel(mx, my, x - 50, y - 30, 0) == \
            xo._mvd_bits_fullpel(mx, my, x - 50, y - 30, 0)
        assert xr._mvd_bits_fullpel(mx, my, x - 50, y - 30, 2) == \
            xo._mvd_bits_fullpel(mx, my, x - 50, y - 30, 2)
        assert xr._mvd_bits(mx, my, x * 3, -y, 0) == xo._mvd_bits(mx, my, x * 3, -y, 0)


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_block(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(9)
    pw, ph = 128, 96
    for comp in (0, 1):
        cs = 1 if comp else 0
        border = 96 >> cs
        padded = rnd_samples(rng, bd, (ph >> cs) + 2 * border, (pw >> cs) + 2 * border)
        for _ in range(40):
            w = int(rng.choice([8, 16, 32])); h = int(rng.choice([8, 16, 32]))
            x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
            y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
            mx, my = int(rng.integers(-3000, 3000)), int(rng.integers(-3000, 3000))
            r = xr.mc_block(bd, comp, x, y, w, h, mx, my, pw, ph, padded, border)
            o = xo.mc_block(bd, comp, x, y, w, h, mx, my, pw, ph, padded, border)
            assert np.array_equal(r, o), (comp, x, y, w, h, mx, my)


def test_transform_tables(libs):
    xo, xr = libs
    for tx in range(1, 6):
        for size in (2, 4, 8, 16, 32, 64):
            r = xr.transform_matrix(tx, size)
            o = xo.transform_matrix(tx, size)
            if r is None:
                assert tx != 1 and size == 2
                continue
            assert np.array_equal(r, o), (tx, size)


TX_DCT2_LOW = 7


def restricted_type(t, size):
    """What the binding passes for TransformType t of a side `size` under
    Restrictions::disable_ext2_transform_high_precision (xvcgpu_types.h,
    XVC_TX_DCT2_LOW): the 6-bit DCT-2 for sizes 4..32, everything else unchanged."""
    return TX_DCT2_LOW if t in (0, 1) and 4 <= size <= 32 else t


def test_low_precision_tables(libs):
    xo, xr = libs
    for size in (4, 8, 16, 32):
        r = xr.transform_matrix(TX_DCT2_LOW, size)
        assert r is not None and np.array_equal(r, xo.transform_matrix(TX_DCT2_LOW, size)), size
        assert int(r[0, 0]) == 64
    for size in (2, 64):
        assert xo.transform_matrix(TX_DCT2_LOW, size) is None


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_transforms_restricted_mode(libs, bd):
    """Restricted mode's transform precision: the reference with the restriction
    flag set against the oracle given XVC_TX_DCT2_LOW where the binding would."""
    xo, xr = libs
    rng = np.random.default_rng(77 + bd)
    for w in [2] + SIZES:
        for h in [2] + SIZES:
            types = [(0, 0), (1, 1)]
            if w >= 4 and h >= 4:
                types += [(3, 5), (1, 5), (5, 1), (0, 2), (4, 0), (2, 2)]
            for tx_hor, tx_ver in types:
                oh, ov = restricted_type(tx_hor, w), restricted_type(tx_ver, h)
                resi = rng.integers(-(1 << bd) + 1, 1 << bd, size=(h, 64)).astype(np.int16)[:, :w]
                r = xr.fwd_transform_restricted(bd, resi, tx_hor, tx_ver)
                o = xo.fwd_transform(bd, resi, oh, ov)
                assert np.array_equal(r, o), ("fwd", w, h, tx_hor, tx_ver)
                if (oh, ov) != (tx_hor, tx_ver) and w * h >= 16:   # it is not the default path
                    assert not np.array_equal(r, xr.fwd_transform(bd, resi, tx_hor, tx_ver))
                coeff = (r.astype(np.int32) // 16 * 16).astype(np.int16)
                ri = xr.inv_transform_restricted(bd, coeff, tx_hor, tx_ver)
                oi = xo.inv_transform(bd, coeff, oh, ov)
                assert np.array_equal(ri, oi), ("inv", w, h, tx_hor, tx_ver)
                dc = np.zeros_like(coeff)
                dc[0, 0] = coeff[0, 0] | 16
                assert np.array_equal(xr.inv_transform_restricted(bd, dc, tx_hor, tx_ver, 0, 1),
                                      xo.inv_transform(bd, dc, oh, ov, 0, 1)), ("dc", w, h)
    # the 4x4 DST of intra luma keeps its own (always 6-bit) path
    resi = rng.integers(-(1 << bd) + 1, 1 << bd, size=(4, 4)).astype(np.int16)
    r = xr.fwd_transform_restricted(bd, resi, 0, 0, 1)
    assert np.array_equal(r, xo.fwd_transform(bd, resi, TX_DCT2_LOW, TX_DCT2_LOW, 1))
    assert np.array_equal(xr.inv_transform_restricted(bd, r, 0, 0, 1),
                          xo.inv_transform(bd, r, TX_DCT2_LOW, TX_DCT2_LOW, 1))


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_transforms(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(21 + bd)
    for w in [2] + SIZES:
        for h in [2] + SIZES:
            types = [(0, 0), (1, 1)]
            if w >= 4 and h >= 4:
                types += [(3, 5), (5, 3), (5, 5), (3, 3), (2, 4), (4, 2), (1, 5), (2, 2), (4, 4)]
            for tx_hor, tx_ver in types:
                resi = rng.integers(-(1 << bd) + 1, 1 << bd, size=(h, 64)).astype(np.int16)[:, :w]
                r = xr.fwd_transform(bd, resi, tx_hor, tx_ver)
                o = xo.fwd_transform(bd, resi, tx_hor, tx_ver)
                assert np.array_equal(r, o), ("fwd", w, h, tx_hor, tx_ver)
                # inverse on plausible dequantised coefficients
                coeff = (r.astype(np.int32) // 16 * 16).astype(np.int16)
                ri = xr.inv_transform(bd, coeff, tx_hor, tx_ver)
                oi = xo.inv_transform(bd, coeff, tx_hor, tx_ver)
                assert np.array_equal(ri, oi), ("inv", w, h, tx_hor, tx_ver)
                # inverse on extreme coefficients (exercises clipping)
                big = rng.integers(-32768, 32767, size=(h, 64)).astype(np.int16)[:, :w]
                assert np.array_equal(xr.inv_transform(bd, big, tx_hor, tx_ver),
                                      xo.inv_transform(bd, big, tx_hor, tx_ver))
            # dc-only shortcut
            dc = np.zeros((h, w), np.int16)
            dc[0, 0] = rng.integers(-3000, 3000)
            assert np.array_equal(xr.inv_transform(bd, dc, 0, 0, 0, 1),
                                  xo.inv_transform(bd, dc, 0, 0, 0, 1))
    # 4x4 DST
    resi = rng.integers(-(1 << bd) + 1, 1 << bd, size=(4, 4)).astype(np.int16)
    r = xr.fwd_transform(bd, resi, 0, 0, 1)
    assert np.array_equal(r, xo.fwd_transform(bd, resi, 0, 0, 1))
    assert np.array_equal(xr.inv_transform(bd, r, 0, 0, 1), xo.inv_transform(bd, r, 0, 0, 1))
    # transform skip
    for (w, h) in [(2, 2), (4, 2), (2, 4), (4, 4)]:
        resi = rng.integers(-(1 << bd) + 1, 1 << bd, size=(h, w)).astype(np.int16)
        r = xr.fwd_transform_skip(bd, resi)
        assert np.array_equal(r, xo.fwd_transform_skip(bd, resi))
        assert np.array_equal(xr.inv_transform_skip(bd, r), xo.inv_transform_skip(bd, r))


@pytest.mark.parametrize("bd", [8, 10])
def test_quant_dequant(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(31)
    for w in [2] + SIZES:
        for h in [2] + SIZES:
            for qp in (0, 7, 22, 27, 32, 37, 51):
                c = (rng.standard_normal((h, w)) * 600).astype(np.int16)
                for intra in (0, 1):
                    rl, rn = xr.quant_fast(bd, qp, intra, c)
                    ol_, on = xo.quant_fast(bd, qp, intra, c)
                    assert rn == on and np.array_equal(rl, ol_), (w, h, qp, intra)
                lv = (rng.standard_normal((h, w)) * 20).astype(np.int16)
                assert np.array_equal(xr.dequant(bd, qp, lv), xo.dequant(bd, qp, lv)), (w, h, qp)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bipred", [0, 1])
def test_deblock(libs, bd, bipred):
    xo, xr = libs
    rng = np.random.default_rng(41 + bd + bipred)
    total_changed = 0
    for (pw, ph) in [(64, 64), (136, 72), (200, 136)]:
        for trial in range(3):
            parts = random_partition(rng, pw, ph)
            l0 = [8, 0, 16][:2 + trial % 2]
            l1 = [16, 8]
            cus, cmap = make_cus(rng, parts, bipred, l0, l1, pw, ph)
            borders = [96, 48, 48]
            planes = []
            for c in range(3):
                w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
                # blocky content so that filters trigger
                base = rng.integers(0, 1 << bd, size=((h + 7) // 8, (w + 7) // 8))
                p = np.kron(base, np.ones((8, 8), np.int64))[:h, :w]
                amp = [2, 6, 30][trial]
                p = np.clip(p // [16, 4, 1][trial] + (1 << (bd - 1)) +
                            rng.integers(-amp, amp + 1, size=(h, w)), 0, (1 << bd) - 1)
                full = np.zeros((h + 2 * borders[c], w + 2 * borders[c]), np.uint16)
                full[borders[c]:borders[c] + h, borders[c]:borders[c] + w] = p
                planes.append(full)
            pr = [p.copy() for p in planes]
            po = [p.copy() for p in planes]
            beta, tc = [(0, 0), (2, -2), (-4, 4)][trial]
            xr.deblock(bd, pw, ph, bipred, beta, tc, 4, cus, cmap, pr, borders, l0, l1)
            xo.deblock(bd, pw, ph, bipred, beta, tc, 4, cus, cmap, po, borders)
            changed = 0
            for c in range(3):
                b = borders[c]
                w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
                assert np.array_equal(pr[c][b:b + h, b:b + w], po[c][b:b + h, b:b + w]), \
                    (pw, ph, trial, c)
                changed += int((pr[c] != planes[c]).sum())
            total_changed += changed
    assert total_changed > 0     # the filters did change samples


def test_pad_border(libs):
    xo, xr = libs
    rng = np.random.default_rng(51)
    for (w, h) in [(64, 64), (136, 72)]:
        borders = [80, 40, 40]
        pr, po = [], []
        for c in range(3):
            cw, ch = (w, h) if c == 0 else (w // 2, h // 2)
            full = rng.integers(0, 1024, size=(ch + 2 * borders[c], cw + 2 * borders[c]),
                                dtype=np.uint16)
            pr.append(full.copy())
            po.append(full.copy())
        xr.pad_border(w, h, pr, borders)
        xo.pad_border(w, h, po, borders)
        for c in range(3):
            assert np.array_equal(pr[c], po[c])


@pytest.mark.parametrize("bd", [8, 10])
def test_tz_and_subpel_search(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(61 + bd)
    pw, ph, border = 192, 128, 96
    n_grid = 0
    for motion in [(3, -2), (0, 0), (-17, 9), (40, 26)]:
        orig, ref = make_pics(rng, bd, pw, ph, border, motion)
        for _ in range(14):
            w = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([8, 16, 32, 64]))
            if w * h < 64:
                continue
            x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
            y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
            blk = ol.MeBlock()
            blk.x, blk.y, blk.w, blk.h = x, y, w, h
            blk.depth_nonzero = int(rng.integers(0, 2))
            blk.fullpel_mv = int(rng.integers(0, 4) == 0)
            blk.mvp_x = int(rng.integers(-200, 200))
            blk.mvp_y = int(rng.integers(-200, 200))
            blk.prev_x = int(rng.integers(-20, 20))
            blk.prev_y = int(rng.integers(-20, 20))
            blk.lambda16 = int(rng.choice([120000, 498000, 1500000]))
            blk.search_range = int(rng.choice([96, 96, 128, 256]))
            (rmv, _) = xr.tz_search(bd, blk, pw, ph, orig, ref, border)
            (omv, ocost) = xo.tz_search(bd, blk, pw, ph, orig, ref, border)
            assert rmv == omv, (motion, x, y, w, h, rmv, omv)
            blk.fullpel_mv = 0
            rs_, rd = xr.subpel_search(bd, blk, pw, ph, orig, ref, border, omv)
            os_, od = xo.subpel_search(bd, blk, pw, ph, orig, ref, border, omv)
            assert rs_ == os_ and rd == od, (motion, x, y, w, h, rs_, os_, rd, od)
            n_grid += 1
    assert n_grid > 30


@pytest.mark.parametrize("bd", [8, 10])
def test_full_search(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(71)
    pw, ph, border = 128, 96, 96
    orig, ref = make_pics(rng, bd, pw, ph, border, (2, 1))
    for _ in range(12):
        w = int(rng.choice([8, 16, 32])); h = int(rng.choice([8, 16, 32]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        o = orig[border + y:border + y + h, border + x:border + x + w].astype(np.int32)
        p = ref[border + y + 1:border + y + 1 + h, border + x:border + x + w].astype(np.int32)
        target = np.zeros((h, 64), np.int16)
        target[:, :w] = 2 * o - p
        mvp = (int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))
        mn, mx = xo.min_max_mv(x, y, pw, ph, mvp[0], mvp[1], 4)
        a = xr.full_search(bd, x, y, w, h, 0, mvp, 498000, mn, mx, target[:, :w], ref, border, pw, ph)
        b = xo.full_search(bd, x, y, w, h, 0, mvp, 498000, mn, mx, target[:, :w], ref, border, pw, ph)
        assert a == b


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_bipred_block(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(83 + bd)
    pw, ph, border = 128, 96, 96
    _, ref0 = make_pics(rng, bd, pw, ph, border, (2, 1))
    _, ref1 = make_pics(rng, bd, pw, ph, border, (-3, 2))
    c0 = np.ascontiguousarray(ref0[::2, ::2])
    c1 = np.ascontiguousarray(ref1[::2, ::2])
    for i in range(40):
        w = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([8, 16, 32, 64]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        big = i % 5 == 0
        lim = 3000 if big else 200
        mv0 = (int(rng.integers(-lim, lim)), int(rng.integers(-lim, lim)))
        mv1 = (int(rng.integers(-lim, lim)), int(rng.integers(-lim, lim)))
        if i % 7 == 0:
            mv1 = (mv1[0] & ~15, mv1[1] & ~15)
        for comp in range(3):
            p0, p1, b = (ref0, ref1, border) if comp == 0 else (c0, c1, border // 2)
            a = xr.mc_bipred_block(bd, comp, x, y, w, h, mv0, mv1, pw, ph, p0, p1, b)
            o = xo.mc_bipred_block(bd, comp, x, y, w, h, mv0, mv1, pw, ph, p0, p1, b)
            assert np.array_equal(a, o), (x, y, w, h, mv0, mv1, comp)


@pytest.mark.parametrize("bd", [8, 10])
def test_bipred_search(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(91 + bd)
    pw, ph, border = 128, 96, 96
    n = 0
    for motion in [(2, 1), (-5, 3)]:
        orig, ref_s = make_pics(rng, bd, pw, ph, border, motion)
        _, ref_o = make_pics(rng, bd, pw, ph, border, (-motion[0], -motion[1]))
        for _ in range(16):
            w = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([8, 16, 32, 64]))
            x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
            y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
            job = ol.BiBlock()
            job.blk.x, job.blk.y, job.blk.w, job.blk.h = x, y, w, h
            job.blk.fullpel_mv = int(rng.integers(0, 4) == 0)
            job.blk.mvp_x = int(rng.integers(-120, 120))
            job.blk.mvp_y = int(rng.integers(-120, 120))
            job.blk.lambda16 = int(rng.choice([120000, 498000, 1500000]))
            job.other_mv_x = int(rng.integers(-100, 100))
            job.other_mv_y = int(rng.integers(-100, 100))
            job.boot_mv_x = motion[0] * 16 + int(rng.integers(-40, 40))
            job.boot_mv_y = motion[1] * 16 + int(rng.integers(-40, 40))
            a = xr.bipred_search(bd, job, pw, ph, orig, ref_o, ref_s, border)
            o = xo.bipred_search(bd, job, pw, ph, orig, ref_o, ref_s, border)
            assert a == o, (x, y, w, h, a, o)
            n += 1
    assert n == 32


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_metric(libs, bd):
    """GetSubpelDist with every metric (T4's per-candidate step)."""
    xo, xr = libs
    rng = np.random.default_rng(97 + bd)
    pw, ph, border = 128, 96, 96
    orig, ref = make_pics(rng, bd, pw, ph, border, (2, -1))
    n = 0
    for i in range(60):
        w = int(rng.choice([8, 16, 32, 64])); h = int(rng.choice([8, 16, 32, 64]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        lim = 3000 if i % 6 == 0 else 150
        mv = (int(rng.integers(-lim, lim)), int(rng.integers(-lim, lim)))
        metric = int(rng.integers(0, 8))
        if metric in (4, 6) and h <= 8:
            metric = 3
        qp, strength = int(rng.integers(20, 45)), int(rng.choice([8, 16]))
        a = xr.mc_metric(bd, metric, qp, strength, x, y, w, h, mv, pw, ph, orig, ref, border)
        o = xo.mc_metric(bd, metric, qp, strength, x, y, w, h, mv, pw, ph, orig, ref, border)
        assert a == o, (metric, x, y, w, h, mv, a, o)
        n += 1
    assert n == 60


def affine_mvs(rng, w, i):
    """Corner MVs of a plausible affine model (plus degenerate / huge cases)."""
    base = (int(rng.integers(-300, 300)), int(rng.integers(-300, 300)))
    if i % 9 == 0:
        return [base, base, (base[0] + 5, base[1] - 3)]        # mv[0] == mv[1]: plain MC
    span = int(rng.choice([1, 3, 8, 40, 200])) if i % 7 else 3000
    mv1 = (base[0] + int(rng.integers(-span, span + 1)), base[1] + int(rng.integers(-span, span + 1)))
    if i % 5 == 0:
        mv2 = base                                            # no vertical variation
    else:
        mv2 = (base[0] + int(rng.integers(-span, span + 1)), base[1] + int(rng.integers(-span, span + 1)))
    return [base, mv1, mv2]


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_affine_block(libs, bd):
    xo, xr = libs
    rng = np.random.default_rng(111 + bd)
    pw, ph, border = 128, 96, 96
    _, ref = make_pics(rng, bd, pw, ph, border, (1, 1))
    cref = np.ascontiguousarray(ref[::2, ::2])
    n_sub = 0
    for i in range(80):
        w = int(rng.choice([16, 32, 64])); h = int(rng.choice([16, 32, 64]))
        x = int(rng.integers(0, (pw - w) // 8 + 1)) * 8
        y = int(rng.integers(0, (ph - h) // 8 + 1)) * 8
        mv3 = affine_mvs(rng, w, i)
        for comp in range(3):
            p, b = (ref, border) if comp == 0 else (cref, border // 2)
            o = xo.mc_affine_block(bd, comp, x, y, w, h, mv3, pw, ph, p, b)
            # The reference's SSE2 chroma filters store 4 columns at a time in
            # high-bit-depth builds (inter_prediction_simd.cc:977-1010) and so
            # differ from its own C kernels on the 2-wide chroma sub-blocks
            # affine MC produces: chroma is pinned against the C kernels, luma
            # against both.
            for simd in ((0, 1) if comp == 0 else (0,)):
                xr._set_simd(simd)
                a = xr.mc_affine_block(bd, comp, x, y, w, h, mv3, pw, ph, p, b)
                assert np.array_equal(a, o), (x, y, w, h, mv3, comp, simd)
        n_sub += mv3[0] != mv3[1]
    xr._set_simd(1)
    assert n_sub > 60


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_quant_fast_sign_hiding(libs, bd):
    """QuantFast as shipped: CoeffSignHideFast on, all three scan orders."""
    xo, xr = libs
    rng = np.random.default_rng(131 + bd)
    n_changed = 0
    for i in range(400):
        w = int(rng.choice([4, 8, 16, 32, 64])); h = int(rng.choice([4, 8, 16, 32, 64]))
        scan = int(rng.integers(0, 3)) if max(w, h) < 16 else 0
        amp = int(rng.choice([30, 300, 3000, 30000]))
        coeff = rng.integers(-amp, amp + 1, size=(h, w)).astype(np.int16)
        if i % 3 == 0:      # sparse, energy near DC like a real TU
            yy, xx = np.mgrid[0:h, 0:w]
            coeff = (coeff / (1 + 0.6 * (xx + yy))).astype(np.int16)
        if i % 11 == 0:
            coeff[rng.integers(0, h), rng.integers(0, w)] = 32767
        qp = int(rng.integers(-6 * (bd - 8), 52))
        intra = int(rng.integers(0, 2))
        a, na = xr.quant_fast2(bd, qp, intra, 1, scan, coeff)
        o, no = xo.quant_fast2(bd, qp, intra, 1, scan, coeff)
        assert na == no and np.array_equal(a, o), (w, h, scan, qp, intra, i)
        plain, _ = xo.quant_fast2(bd, qp, intra, 0, scan, coeff)
        n_changed += not np.array_equal(plain, o)
    assert n_changed > 100   # sign hiding did modify levels in many cases


def test_qp_derivation(libs):
    """Y1: the host-side Qp helpers (chroma qp table, lambda16) against Qp."""
    import ctypes as C
    from xvc_amd import pipeline
    _, xr = libs
    f = xr.dll.xr_qp_info
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32),
                  C.POINTER(C.c_double)]
    for bd in (8, 10):
        for qp in range(0, 64):
            qc, l16, w = C.c_int(), C.c_uint32(), C.c_double()
            f(qp, bd, C.byref(qc), C.byref(l16), C.byref(w))
            assert pipeline.chroma_qp(qp) == qc.value == ol.chroma_qp(qp), qp
            assert pipeline.lambda16_for_qp(qp) == l16.value, qp
            assert w.value == 2.0 ** (-(qc.value - min(qp, 57)) / 3.0)


# Whole composition: the hot-path frame pass executed by the reference's own
# classes (ref_harness.cc xr_frame_pass: TzSearch, InterSearch::SubpelSearch,
# InterPrediction, Forward/InverseTransform, RdoQuant::QuantFast, Quantize,
# DeblockingFilter, PadBorder, ComparePicture) against the oracle's
# restatement, on chains of synthetic pictures (CIF = BASELINE config 0).
@pytest.mark.parametrize("rdoq", [False, True])
@pytest.mark.parametrize("w,h,bd,qp,cu,threads", [
    (352, 288, 10, 32, 16, 1), (352, 288, 8, 22, 8, 4), (136, 72, 10, 37, 16, 2),
    (256, 192, 10, 27, 32, 4), (256, 128, 12, 32, 64, 1)])
def test_frame_pass_composition(libs, w, h, bd, qp, cu, threads, rdoq):
    """rdoq: the quantiser is RdoQuant::QuantRdo (what the reference's encoder
    runs, transform_encoder.cc:230) with the picture-initial context states and
    the host-computed lambda / rd_factor (which xr_frame_pass re-derives from
    its own Qp and asserts equal) - else QuantFast."""
    import oracle_frame
    from xvc_amd import pipeline, synth
    xo, xr = libs
    xr._set_simd(1)
    BL = 128
    clip = synth.SyntheticClip(w, h, bd)
    desc = pipeline.FrameDescriptors(w, h, qp, cu, rdoq=rdoq, bitdepth=bd)

    def padded(planes):
        return [np.ascontiguousarray(np.pad(p, BL >> (c > 0), mode="edge"))
                for c, p in enumerate(planes)]

    ref = padded(clip.frame(0))
    nz = 0
    for n in (1, 2):
        orig = padded(clip.frame(n))
        o_rec, o_res, o_nnz, o_cus, o_ssd = oracle_frame.frame_pass(
            desc, bd, orig, ref, BL, n - 1, lib=xo, threads=threads)
        r_rec, r_res, r_nnz, r_cus, r_ssd = oracle_frame.frame_pass(
            desc, bd, orig, ref, BL, n - 1, lib=xr, threads=threads, reference=True)
        for f in ("fullpel_x", "fullpel_y", "mv_x", "mv_y", "subpel_dist"):
            assert np.array_equal(o_res[f], r_res[f]), (n, f)
        assert np.array_equal(o_nnz, r_nnz)
        assert o_cus.tobytes() == r_cus.tobytes()
        for c in range(3):
            assert np.array_equal(o_rec[c], r_rec[c]), (n, c)
        assert o_ssd == r_ssd
        nz += int(np.count_nonzero(o_nnz))
        ref = o_rec
    assert nz > 0


# ---- whole-picture passes around the hot path (SURVEY 8f N4 + I/O) ----
def _rand_planes(rng, w, h, bd, smooth=True):
    return [rnd_samples(rng, bd, hh, ww, smooth) for ww, hh in
            ((w, h), (w // 2, h // 2), (w // 2, h // 2))]


@pytest.mark.parametrize("in_bd,out_bd", [(8, 8), (8, 10), (10, 10), (10, 12), (8, 12)])
def test_import_picture(libs, in_bd, out_bd):
    import oracle_stats as st
    xo, xr = libs
    rng = np.random.default_rng(700 + in_bd + out_bd)
    # same size (CopyFromBytesFast) and display sizes that are not multiples
    # of 8 (CopyFromBytesWithPadding: internal size rounded up)
    for (iw, ih, ow, oh) in [(64, 48, 64, 48), (50, 30, 56, 32), (36, 22, 40, 24),
                             (350, 286, 352, 288)]:
        data = st.pack_input(_rand_planes(rng, iw, ih, in_bd), in_bd)
        exp = st.xr_import_picture(xr, in_bd, out_bd, iw, ih, ow, oh, data)
        got = st.xo_import_picture(xo, in_bd, out_bd, iw, ih, ow, oh, data)
        for c in range(3):
            assert np.array_equal(got[c], exp[c]), (iw, ih, c)


@pytest.mark.parametrize("bd,out_bd", [(8, 8), (10, 8), (10, 10), (12, 8), (12, 10),
                                       (10, 12), (8, 10)])
@pytest.mark.parametrize("dither", [0, 1])
def test_export_picture(libs, bd, out_bd, dither):
    import oracle_stats as st
    xo, xr = libs
    rng = np.random.default_rng(720 + bd + out_bd)
    for (w, h, dw, dh) in [(64, 48, 64, 48), (56, 32, 50, 30), (352, 288, 350, 286)]:
        planes = _rand_planes(rng, w, h, bd, smooth=bool(dither))
        exp = st.xr_export_picture(xr, bd, out_bd, dither, planes, dw, dh)
        got = st.xo_export_picture(xo, bd, out_bd, dither, planes, dw, dh)
        assert len(exp) == dw * dh * 3 // 2 * (2 if out_bd > 8 else 1)
        assert got == exp, (w, h, dw, dh)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("mode", [0, 1])
def test_picture_crc(libs, bd, mode):
    import oracle_stats as st
    xo, xr = libs
    rng = np.random.default_rng(740 + bd)
    for (w, h) in [(8, 8), (64, 48), (136, 72), (352, 288)]:
        planes = _rand_planes(rng, w, h, bd)
        exp = st.xr_picture_crc(xr, bd, mode, w, h, planes)
        assert len(exp) == (6 if mode else 2)
        assert st.xo_picture_crc(xo, bd, mode, w, h, planes) == exp, (w, h)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_aqp_variance(libs, bd):
    """CalcDeltaQpFromVariance is observable only through the clipped integer
    QP offset: sweep the strength so that every variance lands on several
    different offsets."""
    import oracle_stats as st
    xo, xr = libs
    rng = np.random.default_rng(760 + bd)
    w, h = 192, 128
    hit = set()
    for trial in range(3):
        luma = rnd_samples(rng, bd, h, w, True)
        # flat, textured and noisy regions
        luma[:64, :64] = luma[0, 0]
        luma[64:, 64:128] = rng.integers(0, 1 << bd, (64, 64), dtype=np.uint16)
        luma[:32, 128:] = (luma[:32, 128:] & ~np.uint16(3)) | np.uint16(trial)
        # graded noise: 16x16 tiles with amplitudes 2, 3, 4, 6, 8, 12, ... around mid grey
        amp = [2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96]
        for k, a_ in enumerate(amp):
            a_ = min(a_ << (bd - 8), (1 << (bd - 1)) - 1)
            luma[64:80, 16 * k:16 * k + 16] = (1 << (bd - 1)) + rng.integers(
                -a_, a_ + 1, (16, 16))
        vm = st.xo_variance_map(xo, w, h, luma)
        for ctu in (16, 32, 64):
            for y in range(0, h, ctu):
                for x in range(0, w, ctu):
                    var = st.xo_ctu_variance(xo, w, h, x, y, ctu, vm)
                    for strength in (2, 5, 13, 29, 60):
                        exp = st.xr_aqp_delta_qp(xr, bd, luma, x, y, ctu, strength)
                        assert st.xo_aqp_delta_qp(xo, var, bd, strength) == exp, \
                            (x, y, ctu, strength, var)
                        hit.add(exp)
    assert len(hit) >= 8   # the sweep exercised most of the -3..7 range


@pytest.mark.parametrize("bd", [8, 10])
def test_lic_histogram_distance(libs, bd):
    import oracle_stats as st
    xo, xr = libs
    rng = np.random.default_rng(780 + bd)
    w, h = 64, 48
    thr = int(0.06 * w * h)
    a = rnd_samples(rng, bd, h, w, True)
    a[a == 0] = 1
    seen = set()
    for k in list(range(thr // 2 - 3, thr // 2 + 4)) + [0, 5, w * h // 2]:
        b = a.copy()
        # move k samples to the (otherwise unused) value 0: distance = 2k
        idx = rng.choice(w * h, k, replace=False)
        b.reshape(-1)[idx] = 0
        d = st.xo_histogram_distance(xo, bd, a, b)
        assert d == 2 * k
        allow = st.xo_allow_lic(xo, d, w, h)
        assert allow == st.xr_allow_lic(xr, bd, a, b), k
        seen.add(allow)
    assert seen == {0, 1}
    # general case
    for _ in range(4):
        b = rnd_samples(rng, bd, h, w, True)
        d = st.xo_histogram_distance(xo, bd, a, b)
        ha = np.bincount(a.reshape(-1), minlength=1 << bd).astype(np.int64)
        hb = np.bincount(b.reshape(-1), minlength=1 << bd).astype(np.int64)
        assert d == int(np.abs(ha - hb).sum())
        assert st.xo_allow_lic(xo, d, w, h) == st.xr_allow_lic(xr, bd, a, b)


# ---- intra prediction + SATD mode pre-selection ----
@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("comp", [0, 1])
def test_intra_prediction(libs, bd, comp):
    """IntraPrediction::ComputeRefSamples / FilterRefSamples / Predict for all
    67 modes on blocks with every neighbour configuration."""
    import oracle_intra as oi
    xo, xr = libs
    rng = np.random.default_rng(800 + bd + comp)
    pw, ph = (160, 128) if comp == 0 else (80, 64)
    rec = rnd_samples(rng, bd, ph, pw, True)
    sizes = (4, 8, 16, 32, 64) if comp == 0 else (2, 4, 8, 16, 32)
    jobs = oi.random_jobs(rng, pw, ph, comp, 60, sizes)
    n = 0
    for j in jobs:
        for mode in ([0, 1, 2, 18, 34, 50, 66] + list(rng.integers(2, 67, 10))):
            j["mode"] = mode
            exp = oi.pred_block(xr, "xr", bd, j, rec, pw << (comp > 0), ph << (comp > 0))
            got = oi.pred_block(xo, "xo", bd, j, rec, pw, ph)
            assert np.array_equal(got, exp), (j, mode)
            n += 1
    assert n == 60 * 17


def test_intra_prediction_all_modes_all_sizes(libs):
    import oracle_intra as oi
    xo, xr = libs
    rng = np.random.default_rng(830)
    bd, pw, ph = 10, 192, 192
    rec = rnd_samples(rng, bd, ph, pw, False)
    for w in (4, 8, 16, 32, 64):
        for h in (4, 8, 16, 32, 64):
            j = np.zeros(1, oi.INTRA_DTYPE)[0]
            j["x"], j["y"], j["w"], j["h"] = 64, 64, w, h
            j["neighbors"], j["above_right"], j["below_left"] = 7, h, w
            for mode in range(67):
                j["mode"] = mode
                assert np.array_equal(oi.pred_block(xo, "xo", bd, j, rec, pw, ph),
                                      oi.pred_block(xr, "xr", bd, j, rec, pw, ph)), (w, h, mode)


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_satd_modes(libs, bd):
    import oracle_intra as oi
    xo, xr = libs
    rng = np.random.default_rng(840 + bd)
    pw, ph = 160, 128
    orig, rec = make_pics(rng, bd, pw, ph, 0, motion=(1, 0), noise=6)
    for j in oi.random_jobs(rng, pw, ph, 0, 40):
        exp = oi.satd_modes(xr, "xr", bd, j, orig, rec)
        got = oi.satd_modes(xo, "xo", bd, j, orig, rec)
        assert np.array_equal(got, exp), j
        if j["neighbors"] == 7:
            assert len(set(exp.tolist())) > 8   # the modes really differ


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_intra_lm_chroma(libs, bd):
    """IntraPrediction::PredLmChroma: luma down-scaling, the integer
    least-squares model and its application, for blocks at the picture edges
    and inside, square and not, on correlated, flat and noisy content."""
    import oracle_intra as oi
    xo, xr = libs
    rng = np.random.default_rng(860 + bd)
    pw, ph = 160, 128
    n = 0
    for content in range(4):
        luma = rnd_samples(rng, bd, ph, pw, content != 3)
        base = luma[0::2, 0::2].astype(np.int64)
        mx = (1 << bd) - 1
        if content == 0:      # chroma = a line through luma + noise
            u = np.clip(base * 3 // 4 + 40 + rng.integers(-6, 7, base.shape), 0, mx)
            v = np.clip(mx - base // 2 + rng.integers(-3, 4, base.shape), 0, mx)
        elif content == 1:    # flat chroma
            u = np.full_like(base, 1 << (bd - 1))
            v = np.full_like(base, 17)
        elif content == 2:    # flat luma neighbourhoods
            luma[:] = 300 % mx
            u = rng.integers(0, mx + 1, base.shape)
            v = np.clip(base + rng.integers(-20, 21, base.shape), 0, mx)
        else:
            u = rng.integers(0, mx + 1, base.shape)
            v = rng.integers(0, mx + 1, base.shape)
        planes = [luma, np.ascontiguousarray(u.astype(np.uint16)),
                  np.ascontiguousarray(v.astype(np.uint16))]
        for _ in range(30):
            w, h = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
            x = int(rng.integers(0, (pw // 2 - w) // 2 + 1)) * 2
            y = int(rng.integers(0, (ph // 2 - h) // 2 + 1)) * 2
            if rng.random() < 0.25:
                x = 0
            if rng.random() < 0.25:
                y = 0
            for comp in (1, 2):
                exp = oi.lm_chroma(xr, "xr", bd, comp, x, y, w, h, planes)
                got = oi.lm_chroma(xo, "xo", bd, comp, x, y, w, h, planes)
                assert np.array_equal(got, exp), (content, comp, x, y, w, h)
                n += 1
    assert n == 240


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_lic_block(libs, bd):
    """InterPrediction::MotionCompensationMv with local illumination
    compensation (LocalIlluminationComp / DeriveLicParams)."""
    import oracle_lic as ol_
    BL, BC = 128, 64
    xo, xr = libs
    xr._set_simd(0)     # 2-wide / narrow chroma: pin against the C kernels
    rng = np.random.default_rng(880 + bd)
    pw, ph = 192, 128
    n = 0
    for content in range(3):
        cur, ref = make_pics(rng, bd, pw, ph, BL, motion=(2, 1), noise=3)
        gain = [1.0, 0.8, 1.3][content]
        mx = (1 << bd) - 1
        rec_y = np.clip(cur[BL:BL + ph, BL:BL + pw].astype(np.float64) * gain + 9 * content,
                        0, mx).astype(np.uint16)
        chroma_ref = [rnd_samples(rng, bd, ph // 2 + 2 * BC, pw // 2 + 2 * BC, True)
                      for _ in range(2)]
        rec_c = [np.clip(c[BC:BC + ph // 2, BC:BC + pw // 2].astype(np.int64) + 12, 0, mx)
                 .astype(np.uint16) for c in chroma_ref]
        ref_planes = [np.ascontiguousarray(ref)] + chroma_ref
        rec_planes = [np.ascontiguousarray(rec_y)] + [np.ascontiguousarray(c) for c in rec_c]
        for j, above, left in ol_.random_jobs(rng, pw, ph, 60):
            exp = ol_.xr_mc_lic(xr, bd, j, above, left, pw, ph, ref_planes, [BL, BC, BC],
                                rec_planes)
            got = ol_.xo_mc_lic(xo, bd, j, pw, ph, ref_planes, [BL, BC, BC], rec_planes)
            assert np.array_equal(got, exp), (content, j)
            n += 1
    xr._set_simd(1)
    assert n == 180


# ---- T5: affine motion estimation ---------------------------------------------
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_affine_gradient_search(libs, bd):
    """InterSearch::AffineGradientSearch: Sobel gradients, the 4x5 normal
    equations, elimination, lround - including flat and extreme inputs."""
    import oracle_affine_me as oa
    xo, xr = libs
    rng = np.random.default_rng(1500 + bd)
    mx = (1 << bd) - 1
    nonzero = 0
    for i in range(150):
        w, h = int(rng.choice([16, 32, 64])), int(rng.choice([16, 32, 64]))
        kind = i % 6
        if kind == 0:
            pred = rng.integers(0, mx + 1, size=(h, w))
        elif kind == 1:
            pred = np.full((h, w), int(rng.integers(0, mx + 1)))   # flat: singular system
        elif kind == 2:
            pred = np.tile(rng.integers(0, mx + 1, size=(1, w)), (h, 1))   # no vertical gradient
        elif kind == 3:
            pred = ((np.indices((h, w)).sum(0) % 2) * mx)          # extreme checkerboard
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            pred = (np.sin(xx / 5.0 + i) * np.cos(yy / 7.0) * 0.4 + 0.5) * mx + \
                rng.integers(-3, 4, size=(h, w))
        pred = np.clip(pred, 0, mx).astype(np.uint16)
        amp = int(rng.choice([2, 30, mx]))
        err = rng.integers(-amp, amp + 1, size=(h, w)).astype(np.int16)
        if kind == 4:   # the error a small shift would produce
            err = (np.roll(pred.astype(np.int32), 1, axis=1) - pred).astype(np.int16)
        a = oa.gradient_search(xo, bd, pred, err)
        b = oa.gradient_search(xr, bd, pred, err)
        assert a == b, (i, w, h, kind)
        nonzero += any(a)
    assert nonzero > 25


@pytest.mark.parametrize("bd", [8, 10])
def test_affine_me(libs, bd):
    """InterSearch::MotionEstAffine on zooming / rotating content: uni-pred and
    the bi-pred refinement search (target 2 * orig - other prediction)."""
    import oracle_affine_me as oa
    xo, xr = libs
    rng = np.random.default_rng(1600 + bd)
    pw, ph, border = 192, 128, 128
    moved = iters = boot = nbi = 0
    for (zoom, rot, shift) in [(1.0, 0.0, (1.5, -0.75)), (1.02, 0.0, (0, 0)),
                               (1.0, 0.015, (0.5, 0.5)), (0.985, -0.01, (-2.0, 1.0))]:
        orig, ref = oa.warped_pics(rng, bd, pw, ph, border, zoom, rot, shift)
        _, other = oa.warped_pics(rng, bd, pw, ph, border, 2 - zoom, -rot,
                                  (-shift[0], -shift[1]))
        blocks = oa.random_blocks(rng, pw, ph, 30, bipred=True)
        for b in blocks:
            exp = oa.affine_me(xr, bd, b, pw, ph, orig, ref, border, other)
            got = oa.affine_me(xo, bd, b, pw, ph, orig, ref, border, other)
            assert np.array_equal(got["mv"], exp["mv"]) and got["dist"] == exp["dist"], b
            moved += not np.array_equal(got["mv"], b["mvp"])
            iters += int(got["iterations"])
            boot += bool(b["flags"]) and np.array_equal(got["mv"], b["bootstrap"])
            nbi += bool(int(b["flags"]) & oa.BIPRED)
    assert moved > 60 and iters > 200 and nbi > 40


# ---- Q2: RdoQuant::QuantRdo + CoeffSignHideRdo ---------------------------------
def test_entropy_bits_table(libs):
    xo, xr = libs
    xo.dll.xo_entropy_bits_table.restype = C.POINTER(C.c_uint32 * 128)
    xr.dll.xr_entropy_bits_table.restype = C.POINTER(C.c_uint32 * 128)
    assert list(xo.dll.xo_entropy_bits_table().contents) == \
        list(xr.dll.xr_entropy_bits_table().contents)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_quant_rdo(libs, bd):
    """The oracle's RDOQ == RdoQuant::QuantRdo for every block shape 2..64 x
    2..64, luma and chroma, the three scans, sign hiding on / off, random and
    freshly initialised context states, a range of qp / lambda."""
    import oracle_rdoq as oq
    xo, xr = libs
    rng = np.random.default_rng(4100 + bd)
    sizes = [2, 4, 8, 16, 32, 64]
    n = 0
    for w in sizes:
        for h in sizes:
            for rep in range(8):
                comp = int(rng.integers(0, 3)) if max(w, h) <= 32 else 0
                intra = bool(rng.integers(0, 2))
                scan = 0
                if intra and (w << (1 if comp else 0)) < 16 and (h << (1 if comp else 0)) < 16:
                    scan = int(rng.integers(0, 3))
                flags = (oq.RDOQ_INTRA_CU if intra else 0) | \
                    (oq.RDOQ_NO_2X2 if rng.integers(0, 4) == 0 else 0)
                qp = int(rng.integers(12, 46))
                lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) * float(rng.uniform(0.5, 2.0))
                ctx = oq.random_contexts(rng) if rep % 2 else \
                    oq.init_contexts(xr, bd, qp, int(rng.integers(0, 3)))
                sign_hide = int(rng.integers(0, 4) != 0)
                src = oq.random_coeffs(rng, w, h, bd, rep, qp)
                e_nnz, e_out, prm, cqp = oq.quant_rdo_reference(
                    xr, bd, qp, lam, comp, scan, sign_hide, ctx, flags, src)
                g_nnz, g_out = oq.quant_rdo_oracle(xo, bd, cqp, comp, scan, sign_hide, ctx, prm,
                                                   src)
                assert g_nnz == e_nnz and np.array_equal(g_out, e_out), \
                    (bd, w, h, comp, scan, sign_hide, flags, qp, rep)
                n += e_nnz > 1
    assert n > 140      # most of the 288 cases code several levels
