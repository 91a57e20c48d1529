"""GPU parity of the whole-picture passes around the hot path (k_stats.h:
sample conversion in / out, CRC, AQP variance statistic, LIC histogram
distance) against the oracle, through the C-ABI.  Bit / byte exact."""
import numpy as np
import pytest

import oracle_lib as ol
import oracle_stats as st
from helpers import rnd_samples

pytestmark = pytest.mark.gpu
BL, BC = 128, 64


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


@pytest.fixture(scope="module")
def xo():
    return ol.Lib("xo")


def rand_planes(rng, w, h, bd, smooth=True):
    return [rnd_samples(rng, bd, hh, ww, smooth) for ww, hh in
            ((w, h), (w // 2, h // 2), (w // 2, h // 2))]


def padded(planes):
    return [np.ascontiguousarray(np.pad(p, BL >> (c > 0), mode="edge"))
            for c, p in enumerate(planes)]


SIZES = [(64, 48, 64, 48), (50, 30, 56, 32), (36, 22, 40, 24), (350, 286, 352, 288),
         (1920, 1080, 1920, 1080), (2046, 1078, 2048, 1080)]


@pytest.mark.parametrize("in_bd,bd", [(8, 8), (8, 10), (10, 10), (10, 12), (8, 12)])
def test_picture_import(gpu, xo, in_bd, bd):
    api, ctx = gpu
    rng = np.random.default_rng(900 + in_bd + bd)
    for (iw, ih, w, h) in SIZES:
        data = st.pack_input(rand_planes(rng, iw, ih, in_bd, smooth=False), in_bd)
        exp = st.xo_import_picture(xo, in_bd, bd, iw, ih, w, h, data)
        P = ctx.picture(w, h, bd)
        ctx.picture_import(P, data, iw, ih, in_bd)
        got = P.download()
        for c in range(3):
            assert np.array_equal(got[c], exp[c]), (iw, ih, c)
        P.destroy()


@pytest.mark.parametrize("bd,out_bd", [(8, 8), (10, 8), (10, 10), (12, 8), (12, 10),
                                       (10, 12), (8, 10)])
@pytest.mark.parametrize("dither", [0, 1])
def test_picture_export(gpu, xo, bd, out_bd, dither):
    api, ctx = gpu
    rng = np.random.default_rng(920 + bd + out_bd)
    for (dw, dh, w, h) in SIZES + [(7680, 4320, 7680, 4320)][:1 if dither and bd == 10 else 0]:
        planes = rand_planes(rng, w, h, bd, smooth=bool(dither))
        exp = st.xo_export_picture(xo, bd, out_bd, dither, planes, dw, dh)
        P = ctx.picture(w, h, bd)
        P.upload(planes)
        got = ctx.picture_export(P, dw, dh, out_bd, dither)
        assert got == exp, (w, h, dw, dh)
        P.destroy()


def test_import_export_round_trip(gpu):
    """8-bit input -> internal 10 bit -> 8-bit output (both down-shift kinds)
    gives the input back: a size-independent property at 2160p."""
    api, ctx = gpu
    rng = np.random.default_rng(5)
    w, h = 3840, 2160
    data = st.pack_input(rand_planes(rng, w, h, 8, smooth=False), 8)
    P = ctx.picture(w, h, 10)
    ctx.picture_import(P, data, w, h, 8)
    assert ctx.picture_export(P, w, h, 8, False) == data
    assert ctx.picture_export(P, w, h, 8, True) == data
    P.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("mode", [0, 1])
def test_picture_crc(gpu, xo, bd, mode):
    api, ctx = gpu
    rng = np.random.default_rng(940 + bd)
    for (w, h) in [(8, 8), (64, 48), (136, 72), (352, 288), (1032, 520), (1920, 1080),
                   (2056, 1096)]:
        planes = rand_planes(rng, w, h, bd, smooth=False)
        P = ctx.picture(w, h, bd)
        P.upload(planes)
        assert ctx.picture_crc(P, mode) == st.xo_picture_crc(xo, bd, mode, w, h, planes), \
            (w, h)
        P.destroy()


def test_picture_crc_is_sensitive(gpu, xo):
    api, ctx = gpu
    rng = np.random.default_rng(7)
    w, h, bd = 352, 288, 10
    planes = rand_planes(rng, w, h, bd)
    P = ctx.picture(w, h, bd)
    P.upload(planes)
    base = ctx.picture_crc(P, 1)
    seen = {base}
    for (c, y, x) in [(0, 0, 0), (0, h - 1, w - 1), (1, 17, 3), (2, h // 2 - 1, w // 2 - 1)]:
        q = [p.copy() for p in planes]
        q[c][y, x] ^= 1
        P.upload(q)
        got = ctx.picture_crc(P, 1)
        assert got == st.xo_picture_crc(xo, bd, 1, w, h, q)
        assert got[2 * c:2 * c + 2] != base[2 * c:2 * c + 2]
        seen.add(got)
    assert len(seen) == 5
    P.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_variance_map(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(960 + bd)
    for (w, h) in [(64, 64), (192, 128), (136, 72), (1920, 1080)]:
        planes = rand_planes(rng, w, h, bd)
        planes[0][: h // 2, : w // 2] = rng.integers(0, 1 << bd, (h // 2, w // 2),
                                                     dtype=np.uint16)
        planes[0][:16, :16] = (1 << bd) - 1    # extremes: all max, max/0 checkerboard
        if w >= 48:
            planes[0][:16, 16:32] = ((np.indices((16, 16)).sum(0) & 1) * ((1 << bd) - 1))
        pp = padded(planes)
        P = ctx.picture(w, h, bd)
        P.upload(pp, BL)
        luma = pp[0][BL:, BL:]           # hang-over blocks read the border
        exp = st.xo_variance_map(xo, w, h, luma)
        for ctu in (16, 32, 64):
            v, cv = ctx.variance_map(P, ctu)
            assert np.array_equal(v, exp), (w, h)
            for cy in range(cv.shape[0]):
                for cx in range(cv.shape[1]):
                    assert int(cv[cy, cx]) == st.xo_ctu_variance(
                        xo, w, h, cx * ctu, cy * ctu, ctu, exp), (w, h, ctu, cx, cy)
        P.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_histogram_distance(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(980 + bd)
    for (w, h) in [(64, 48), (136, 72), (1920, 1080)]:
        a = rand_planes(rng, w, h, bd)
        b = rand_planes(rng, w, h, bd)
        A, B = ctx.picture(w, h, bd), ctx.picture(w, h, bd)
        A.upload(a)
        B.upload(b)
        for _ in range(2):      # the scratch histogram is left cleared
            assert ctx.histogram_distance(A, B) == st.xo_histogram_distance(
                xo, bd, a[0], b[0])
        assert ctx.histogram_distance(A, A) == 0
        b[0][:] = 0             # everything in one bucket
        B.upload(b)
        assert ctx.histogram_distance(A, B) == st.xo_histogram_distance(xo, bd, a[0], b[0])
        A.destroy()
        B.destroy()


def test_stats_error_paths(gpu):
    api, ctx = gpu
    P = ctx.picture(64, 48, 10)
    Q = ctx.picture(64, 64, 10)
    d = ctx.alloc(64 * 48 * 3)
    lib = ctx.lib
    assert lib.xvcgpu_picture_import(ctx.h, P.h_pic, d.ptr, 72, 48, 8) == 10
    assert lib.xvcgpu_picture_import(ctx.h, P.h_pic, d.ptr, 64, 48, 12) == 10
    assert lib.xvcgpu_picture_export(ctx.h, P.h_pic, d.ptr, 64, 50, 8, 0) == 10
    assert lib.xvcgpu_picture_crc(ctx.h, P.h_pic, 2, d.ptr) == 10
    assert lib.xvcgpu_variance_map(ctx.h, P.h_pic, d.ptr, 48, d.ptr) == 10
    assert lib.xvcgpu_histogram_distance(ctx.h, P.h_pic, Q.h_pic, d.ptr) == 10
    d.free()
    P.destroy()
    Q.destroy()
