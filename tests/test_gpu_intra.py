"""GPU parity of intra prediction and the SATD mode pre-selection pass
(k_intra.h) against the oracle, through the C-ABI.  Bit exact."""
import numpy as np
import pytest

import oracle_intra as oi
import oracle_lib as ol
from helpers import make_pics, rnd_samples

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


def upload(ctx, planes, w, h, bd):
    P = ctx.picture(w, h, bd)
    P.upload(planes)
    return P


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_intra_pred_batch(gpu, xo, bd):
    """Random blocks of all sizes / neighbour configurations / modes in all
    three components, one batch per component (blocks may overlap: compare
    job by job in separate launches of non-overlapping subsets)."""
    api, ctx = gpu
    rng = np.random.default_rng(1000 + bd)
    w, h = 320, 256
    planes = [rnd_samples(rng, bd, hh, ww, True)
              for ww, hh in ((w, h), (w // 2, h // 2), (w // 2, h // 2))]
    R = upload(ctx, planes, w, h, bd)
    P = ctx.picture(w, h, bd)
    total = 0
    for comp in (0, 1, 2):
        pw, ph = planes[comp].shape[1], planes[comp].shape[0]
        sizes = (4, 8, 16, 32, 64) if comp == 0 else (2, 4, 8, 16, 32)
        jobs = oi.random_jobs(rng, pw, ph, comp, 160, sizes)
        # greedy split into batches of mutually disjoint blocks
        batches = []
        def box(j):
            return int(j["x"]), int(j["y"]), int(j["x"]) + int(j["w"]), int(j["y"]) + int(j["h"])

        for j in jobs:
            a = box(j)
            for bt in batches:
                if all(a[2] <= b[0] or b[2] <= a[0] or a[3] <= b[1] or b[3] <= a[1]
                       for b in map(box, bt)):
                    bt.append(j)
                    break
            else:
                batches.append([j])
        for bt in batches:
            arr = np.array(bt, oi.INTRA_DTYPE)
            ctx.intra_pred_batch(R, P, arr)
            got = P.download()[comp]
            for j in arr:
                exp = oi.pred_block(xo, "xo", bd, j, planes[comp], pw, ph)
                x0, y0, x1, y1 = box(j)
                assert np.array_equal(got[y0:y1, x0:x1], exp), (comp, j)
                total += 1
    assert total == 480
    R.destroy()
    P.destroy()


def test_intra_pred_all_modes_all_sizes(gpu, xo):
    api, ctx = gpu
    rng = np.random.default_rng(1030)
    bd, w, h = 10, 256, 256
    planes = [rnd_samples(rng, bd, hh, ww, False)
              for ww, hh in ((w, h), (w // 2, h // 2), (w // 2, h // 2))]
    R = upload(ctx, planes, w, h, bd)
    P = ctx.picture(w, h, bd)
    for bw in (4, 8, 16, 32, 64):
        for bh in (4, 8, 16, 32, 64):
            for (nb, ar, bl) in ((7, bh, bw), (7, 0, 0), (6, bh // 2, 0), (5, 0, bw // 2),
                                 (0, 0, 0)):
                for m0 in range(0, 67, 9):
                    jobs = np.zeros(9, oi.INTRA_DTYPE)
                    for k, j in enumerate(jobs):   # a 3x3 grid of disjoint blocks
                        j["x"], j["y"] = 64 * (k % 3) + 8, 64 * (k // 3) + 8
                        j["w"], j["h"], j["mode"] = bw, bh, min(66, m0 + k)
                        j["neighbors"], j["above_right"], j["below_left"] = nb, ar, bl
                        if int(j["x"]) + bw + ar > w or int(j["y"]) + bh + bl > h:
                            j["above_right"], j["below_left"] = 0, 0
                    ctx.intra_pred_batch(R, P, jobs)
                    got = P.download()[0]
                    for j in jobs:
                        exp = oi.pred_block(xo, "xo", bd, j, planes[0], w, h)
                        x, y = int(j["x"]), int(j["y"])
                        assert np.array_equal(got[y:y + bh, x:x + bw], exp), j
    R.destroy()
    P.destroy()


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_satd_batch(gpu, xo, bd):
    api, ctx = gpu
    rng = np.random.default_rng(1040 + bd)
    w, h = 320, 256
    orig, rec = make_pics(rng, bd, w, h, 0, motion=(1, 0), noise=6)
    chroma = np.full((h // 2, w // 2), 1 << (bd - 1), np.uint16)
    O = upload(ctx, [orig, chroma, chroma], w, h, bd)
    R = upload(ctx, [rec, chroma, chroma], w, h, bd)
    for sizes in ((4, 8, 16, 32, 64), (4, 8, 16, 32), (4, 8, 16)):   # the three tile sizes
        jobs = oi.random_jobs(rng, w, h, 0, 120, sizes)
        got = ctx.intra_satd_batch(O, R, jobs)
        assert got.shape == (120, 67)
        for j, g in zip(jobs, got):
            assert np.array_equal(g, oi.satd_modes(xo, "xo", bd, j, orig, rec)), j
    O.destroy()
    R.destroy()


def test_intra_picture_of_ctus(gpu, xo):
    """1080p: every 16x16 block of the picture as an intra job against the
    previous reconstruction (full neighbours away from the edges): the table
    of 8160 x 67 distortions equals the oracle's on a sample of rows, and the
    best mode's prediction equals the original where the picture is flat."""
    api, ctx = gpu
    from xvc_amd import synth
    w, h, bd = 1920, 1080, 10
    clip = synth.SyntheticClip(w, h, bd)
    f0, f1 = clip.frame(0), clip.frame(1)
    O, R = upload(ctx, f1, w, h, bd), upload(ctx, f0, w, h, bd)
    jobs = []
    for y in range(0, h - 8, 16):
        for x in range(0, w, 16):
            bh = min(16, h - y)
            nb = (oi.HAS_LEFT if x else 0) | (oi.HAS_ABOVE if y else 0) | \
                (oi.HAS_ABOVE_LEFT if x and y else 0)
            ar = min(bh, w - x - 16) if y else 0
            bl = 0          # raster coding order: below-left not yet coded
            jobs.append((x, y, 16, bh, 0, 0, nb, ar, bl, 0))
    jobs = np.array(jobs, oi.INTRA_DTYPE)
    got = ctx.intra_satd_batch(O, R, jobs)
    assert got.shape == (len(jobs), 67)
    rng = np.random.default_rng(3)
    for k in rng.choice(len(jobs), 150, replace=False):
        assert np.array_equal(got[k], oi.satd_modes(xo, "xo", bd, jobs[k], f1[0], f0[0])), k
    O.destroy()
    R.destroy()


def test_intra_error_paths(gpu):
    api, ctx = gpu
    P, Q = ctx.picture(64, 48, 10), ctx.picture(64, 64, 10)
    d = ctx.alloc(1024)
    lib = ctx.lib
    assert lib.xvcgpu_intra_pred_batch(ctx.h, P.h_pic, Q.h_pic, d.ptr, 1) == 10
    assert lib.xvcgpu_intra_satd_batch(ctx.h, P.h_pic, Q.h_pic, d.ptr, 1, d.ptr, 64) == 10
    assert lib.xvcgpu_intra_satd_batch(ctx.h, P.h_pic, P.h_pic, None, 1, d.ptr, 64) == 10
    assert lib.xvcgpu_intra_satd_batch(ctx.h, P.h_pic, P.h_pic, d.ptr, 1, d.ptr, 128) == 10
    assert lib.xvcgpu_intra_pred_batch(ctx.h, P.h_pic, P.h_pic, None, 0) == 0
    # decoder form of the fused entry point needs the levels
    assert lib.xvcgpu_intra_recon_batch(ctx.h, None, P.h_pic, d.ptr, d.ptr, 1, None, None,
                                        None) == 10
    assert lib.xvcgpu_intra_recon_batch(ctx.h, Q.h_pic, P.h_pic, d.ptr, d.ptr, 1, None, None,
                                        None) == 10
    d.free()
    P.destroy()
    Q.destroy()


@pytest.mark.parametrize("w,h,bd,qp,cu,fused", [
    (352, 288, 10, 32, 16, True), (352, 288, 10, 32, 16, False), (136, 72, 8, 27, 8, True),
    (136, 72, 12, 27, 8, False), (256, 192, 10, 37, 32, False), (512, 320, 10, 32, 64, False),
    (1920, 1080, 10, 32, 16, True)])
def test_intra_picture_pass(gpu, xo, w, h, bd, qp, cu, fused):
    """An all-intra picture on the device, wave by wave (anti-diagonals of the
    CU raster), against the CU-by-CU oracle composition: chosen modes, levels,
    coefficient counts and the reconstruction; then the decoder's side from the
    same syntax (no host round trips) reproduces the reconstruction."""
    import oracle_intra_picture
    from xvc_amd import pipeline, synth
    api, ctx = gpu
    clip = synth.SyntheticClip(w, h, bd)
    orig = clip.frame(3)
    O = upload(ctx, orig, w, h, bd)
    R, D = ctx.picture(w, h, bd), ctx.picture(w, h, bd)
    ip = pipeline.IntraPicturePass(ctx, w, h, bd, qp, cu, fused=fused)
    if fused:   # the host fold and the device fold choose the same modes
        ip.encode(O, R, host_select=True)
        host_modes = ip.results()[0]
    ip.encode(O, R)
    modes, levels, nnz = ip.results()
    if fused:
        assert np.array_equal(modes, host_modes)
    e_desc = pipeline.IntraPictureDescriptors(w, h, qp, cu)
    e_rec, e_modes, e_levels, e_nnz = oracle_intra_picture.run(xo, e_desc, bd, orig)
    assert np.array_equal(modes, e_modes)
    assert np.array_equal(nnz, e_nnz) and np.array_equal(levels, e_levels)
    got = R.download()
    for c in range(3):
        assert np.array_equal(got[c], e_rec[c]), c
    assert len(set(modes.tolist())) > 5 and np.count_nonzero(nnz) > 0
    # decoder side
    ip.load(modes, levels, nnz)
    ip.decode(D)
    ctx.sync()
    dec = D.download()
    for c in range(3):
        assert np.array_equal(dec[c], e_rec[c]), c
    ip.destroy()
    for p in (O, R, D):
        p.destroy()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_intra_lm_chroma(gpu, xo, bd):
    """LM chroma jobs (mode 67) of the prediction batch against the oracle."""
    api, ctx = gpu
    rng = np.random.default_rng(1060 + bd)
    w, h = 320, 256
    mx = (1 << bd) - 1
    luma = rnd_samples(rng, bd, h, w, True)
    base = luma[0::2, 0::2].astype(np.int64)
    u = np.clip(base * 3 // 4 + 40 + rng.integers(-6, 7, base.shape), 0, mx).astype(np.uint16)
    v = rng.integers(0, mx + 1, base.shape).astype(np.uint16)
    v[:40, :60] = 99 % mx
    planes = [luma, np.ascontiguousarray(u), np.ascontiguousarray(v)]
    R = upload(ctx, planes, w, h, bd)
    P = ctx.picture(w, h, bd)
    total = 0
    for trial in range(6):
        jobs = []
        for gy in range(4):          # disjoint blocks on a 40 x 32 grid (chroma units)
            for gx in range(4):
                bw, bh = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
                x = gx * 40 + (0 if trial % 2 == 0 and gx == 0 else 2 * int(rng.integers(0, 4)))
                y = gy * 32 + (0 if trial % 3 == 0 and gy == 0 else 0)
                for comp in (1, 2):
                    jobs.append((x, y, bw, bh, comp, 67, 0, 0, 0, 0))
        jobs = np.array(jobs, oi.INTRA_DTYPE)
        ctx.intra_pred_batch(R, P, jobs)
        got = P.download()
        for j in jobs:
            x, y, bw, bh, comp = (int(j[k]) for k in ("x", "y", "w", "h", "comp"))
            exp = oi.lm_chroma(xo, "xo", bd, comp, x, y, bw, bh, planes)
            assert np.array_equal(got[comp][y:y + bh, x:x + bw], exp), j
            total += 1
    assert total == 6 * 32
    R.destroy()
    P.destroy()
