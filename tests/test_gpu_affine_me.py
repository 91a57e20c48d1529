"""GPU parity of the affine motion estimation (k_affine_me.h: T5,
InterSearch::MotionEstAffine + AffineGradientSearch) against the oracle,
through the C-ABI.  Bit exact, including the float / double arithmetic of the
reference (see the kernel header for why the sums are order independent)."""
import numpy as np
import pytest

import oracle_affine_me as oa
import oracle_lib as ol

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


def upload(ctx, luma_pad, pw, ph, bd):
    chroma = np.full((ph // 2 + 2 * BC, pw // 2 + 2 * BC), 1 << (bd - 1), np.uint16)
    pic = ctx.picture(pw, ph, bd)
    pic.upload([np.ascontiguousarray(luma_pad), chroma, chroma], BL)
    return pic


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_affine_me_batch(gpu, xo, bd):
    api, ctx = gpu
    assert api.AFFINE_ME_DTYPE == oa.BLOCK_DTYPE and api.AFFINE_ME_RESULT_DTYPE == oa.RESULT_DTYPE
    rng = np.random.default_rng(5200 + bd)
    pw, ph = 192, 128
    moved = iters = best_boot = nbi = 0
    for (zoom, rot, shift) in [(1.0, 0.0, (1.5, -0.75)), (1.02, 0.0, (0, 0)),
                               (1.0, 0.015, (0.5, 0.5)), (0.985, -0.01, (-2.0, 1.0)),
                               (1.0, 0.0, (0, 0))]:
        orig, ref = oa.warped_pics(rng, bd, pw, ph, BL, zoom, rot, shift)
        _, other = oa.warped_pics(rng, bd, pw, ph, BL, 2 - zoom, -rot, (-shift[0], -shift[1]))
        O, R = upload(ctx, orig, pw, ph, bd), upload(ctx, ref, pw, ph, bd)
        T = upload(ctx, other, pw, ph, bd)
        blocks = oa.random_blocks(rng, pw, ph, 48, bipred=True)
        nbi += int((blocks["flags"] & oa.BIPRED != 0).sum())
        got = ctx.affine_me_batch(O, R, blocks, T)
        for b, g in zip(blocks, got):
            e = oa.affine_me(xo, bd, b, pw, ph, orig, ref, BL, other)
            assert np.array_equal(g["mv"], e["mv"]) and g["dist"] == e["dist"] and \
                g["iterations"] == e["iterations"], (b, g, e)
            moved += not np.array_equal(g["mv"], b["mvp"])
            iters += int(g["iterations"])
            best_boot += bool(b["flags"]) and np.array_equal(g["mv"], b["bootstrap"])
        O.destroy()
        R.destroy()
        T.destroy()
    assert moved > 150 and iters > 500 and nbi > 100


def test_affine_me_flat_and_extreme(gpu, xo):
    """Singular systems (flat prediction), saturated content and vectors far
    outside the picture."""
    api, ctx = gpu
    bd, pw, ph = 10, 128, 128
    rng = np.random.default_rng(5300)
    mx = (1 << bd) - 1
    H, W = ph + 2 * BL, pw + 2 * BL
    cases = [
        (np.full((H, W), 500, np.uint16), np.full((H, W), 500, np.uint16)),
        (np.full((H, W), 500, np.uint16), np.full((H, W), 900, np.uint16)),
        (((np.indices((H, W)).sum(0) % 2) * mx).astype(np.uint16),
         ((np.indices((H, W)).sum(0) % 2 == 0) * mx).astype(np.uint16)),
        (rng.integers(0, mx + 1, size=(H, W)).astype(np.uint16),
         rng.integers(0, mx + 1, size=(H, W)).astype(np.uint16)),
    ]
    for orig, ref in cases:
        O, R = upload(ctx, orig, pw, ph, bd), upload(ctx, ref, pw, ph, bd)
        blocks = oa.random_blocks(rng, pw, ph, 24)
        blocks["mvp"][::5] *= 40
        got = ctx.affine_me_batch(O, R, blocks)
        for b, g in zip(blocks, got):
            e = oa.affine_me(xo, bd, b, pw, ph, orig, ref, BL)
            assert np.array_equal(g["mv"], e["mv"]) and g["dist"] == e["dist"] and \
                g["iterations"] == e["iterations"], (b, g, e)
        O.destroy()
        R.destroy()
    assert len(ctx.affine_me_batch(upload(ctx, cases[0][0], pw, ph, bd),
                                   upload(ctx, cases[0][1], pw, ph, bd),
                                   np.zeros(0, api.AFFINE_ME_DTYPE))) == 0
