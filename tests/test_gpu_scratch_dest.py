"""Scratch destinations for the RD search (xvcgpu_inter_pred_batch_to,
xvcgpu_copy_blocks): the prediction of a candidate CU written to a caller-chosen
slot of a scratch picture must be the block xvcgpu_inter_pred_batch writes at
the CU's own position (itself pinned against the reference through the decoded
streams), for uni- / bi-prediction, affine CUs and all three components."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BL = 80


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def _picture(ctx, rng, w, h):
    planes = [rng.integers(0, 1024, (h + 2 * BL, w + 2 * BL)).astype(np.uint16),
              rng.integers(0, 1024, (h // 2 + BL, w // 2 + BL)).astype(np.uint16),
              rng.integers(0, 1024, (h // 2 + BL, w // 2 + BL)).astype(np.uint16)]
    p = ctx.picture(w, h, 10)
    p.upload(planes, BL)
    return p


@pytest.mark.parametrize("seed", [1, 2])
def test_prediction_to_scratch_equals_prediction_in_place(gpu, seed):
    api, ctx = gpu
    rng = np.random.default_rng(seed)
    W, H = 320, 192
    refs = [_picture(ctx, rng, W, H) for _ in range(3)]
    rec = _picture(ctx, rng, W, H)
    pred = ctx.picture(W, H, 10)
    # non-overlapping CUs: one per 64x64 cell, random size inside it
    cells = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)]
    n = len(cells)
    jobs = np.zeros((n, 3), api.INTER_DTYPE)
    for i, (x, y) in enumerate(cells):
        w = int(rng.choice([8, 16, 32, 64]))
        h = int(rng.choice([8, 16, 32, 64]))
        kind = int(rng.integers(0, 3))          # 0 uni, 1 bi, 2 affine
        affine = kind == 2 and w >= 16 and h >= 16
        lic = (not affine) and rng.integers(0, 3) == 0 and x > 0 and y > 0
        for c in range(3):
            j = jobs[i, c]
            j["x"], j["y"], j["w"], j["h"], j["comp"] = x, y, w, h, c
            j["flags"] = (api.INTER_AFFINE if affine else 0) | (api.INTER_LIC if lic else 0)
        ref = [int(rng.integers(0, 3)), int(rng.integers(0, 3))]
        if kind == 0 or (affine and rng.integers(0, 2)):
            ref[int(rng.integers(0, 2))] = -1
        mv = rng.integers(-200, 200, (2, 3, 2))
        for c in range(3):
            jobs[i, c]["ref"] = ref
            jobs[i, c]["mv"] = mv
            if lic:
                jobs[i, c]["neighbors"] = 3
                jobs[i, c]["above_x"], jobs[i, c]["above_y"] = x, y - 8
                jobs[i, c]["left_x"], jobs[i, c]["left_y"] = x - 8, y
    flat = jobs.reshape(-1)
    ctx.inter_pred_batch(refs, rec, pred, flat)
    want = pred.download()
    # scratch: a picture of another size, slots in shuffled order
    SW, SH = 512, 64 * ((n + 7) // 8)
    scratch = ctx.picture(SW, SH, 10)
    perm = rng.permutation(n)
    dst = np.zeros((n, 3), api.POS_DTYPE)
    dst["x"] = ((perm % 8) * 64)[:, None]
    dst["y"] = ((perm // 8) * 64)[:, None]
    ctx.inter_pred_batch_to(refs, rec, scratch, flat, dst.reshape(-1))
    got = scratch.download()
    # the originals beside: copy the CU's block of `rec` to the slot of a second scratch
    side = ctx.picture(SW, SH, 10)
    cp = np.zeros((n, 3), api.COPY_BLOCK_DTYPE)
    for c in range(3):
        s = 1 if c else 0
        cp["sx"][:, c], cp["sy"][:, c] = jobs["x"][:, c] >> s, jobs["y"][:, c] >> s
        cp["dx"][:, c], cp["dy"][:, c] = dst["x"][:, c] >> s, dst["y"][:, c] >> s
        cp["w"][:, c], cp["h"][:, c] = jobs["w"][:, c] >> s, jobs["h"][:, c] >> s
        cp["comp"][:, c] = c
    ctx.copy_blocks(rec, side, cp.reshape(-1))
    copied = side.download()
    recp = rec.download()
    for i in range(n):
        for c in range(3):
            s = 1 if c else 0
            x, y = int(jobs[i, c]["x"]) >> s, int(jobs[i, c]["y"]) >> s
            w, h = int(jobs[i, c]["w"]) >> s, int(jobs[i, c]["h"]) >> s
            dx, dy = int(dst[i, c]["x"]) >> s, int(dst[i, c]["y"]) >> s
            assert np.array_equal(got[c][dy:dy + h, dx:dx + w], want[c][y:y + h, x:x + w]), (i, c)
            assert np.array_equal(copied[c][dy:dy + h, dx:dx + w], recp[c][y:y + h, x:x + w]), (i, c)
    # nothing outside the slots was written
    mask = np.zeros((SH, SW), bool)
    for i in range(n):
        dx, dy = int(dst[i, 0]["x"]), int(dst[i, 0]["y"])
        mask[dy:dy + int(jobs[i, 0]["h"]), dx:dx + int(jobs[i, 0]["w"])] = True
    assert not got[0][~mask].any() and not copied[0][~mask].any()
    for p in refs + [rec, pred, scratch, side]:
        p.destroy()


def test_scratch_destination_arguments(gpu):
    api, ctx = gpu
    a = ctx.picture(64, 64, 10)
    b = ctx.picture(128, 64, 8)
    jobs = np.zeros(1, api.INTER_DTYPE)
    arr = (api._vp * 1)(a.h_pic)
    d = ctx.buffer(jobs)
    # a destination list is required, bit depths must agree
    assert ctx.lib.xvcgpu_inter_pred_batch_to(ctx.h, arr, 1, a.h_pic, a.h_pic, d.ptr, None, 1) != 0
    dd = ctx.buffer(np.zeros(1, api.POS_DTYPE))
    assert ctx.lib.xvcgpu_inter_pred_batch_to(ctx.h, arr, 1, a.h_pic, b.h_pic, d.ptr, dd.ptr, 1) != 0
    assert ctx.lib.xvcgpu_copy_blocks(ctx.h, a.h_pic, b.h_pic, d.ptr, 1) != 0
    assert ctx.lib.xvcgpu_copy_blocks(ctx.h, a.h_pic, a.h_pic, None, 0) == 0
    d.free()
    dd.free()
    a.destroy()
    b.destroy()


def test_copy_blocks_any_width(gpu):
    """Widths that are not powers of two take the row loop."""
    api, ctx = gpu
    rng = np.random.default_rng(9)
    src = _picture(ctx, rng, 128, 64)
    dst = ctx.picture(192, 96, 10)
    jobs = np.zeros(3, api.COPY_BLOCK_DTYPE)
    jobs["sx"], jobs["sy"] = [3, 10, 0], [5, 0, 2]
    jobs["dx"], jobs["dy"] = [7, 60, 1], [9, 40, 3]
    jobs["w"], jobs["h"] = [12, 24, 6], [5, 7, 3]
    jobs["comp"] = [0, 0, 1]
    ctx.copy_blocks(src, dst, jobs)
    a, b = src.download(), dst.download()
    for j in jobs:
        c = int(j["comp"])
        assert np.array_equal(b[c][j["dy"]:j["dy"] + j["h"], j["dx"]:j["dx"] + j["w"]],
                              a[c][j["sy"]:j["sy"] + j["h"], j["sx"]:j["sx"] + j["w"]])
    src.destroy()
    dst.destroy()


def test_page_locked_upload(gpu):
    """xvcgpu_host_alloc + xvcgpu_memcpy_h2d_async + an event: the bytes arrive,
    ordered with the stream."""
    import ctypes as C
    api, ctx = gpu
    n = 1 << 20
    hp = C.c_void_p()
    assert ctx.lib.xvcgpu_host_alloc(ctx.h, n, C.byref(hp)) == 0 and hp.value
    host = np.ctypeslib.as_array((C.c_uint8 * n).from_address(hp.value))
    host[:] = np.arange(n, dtype=np.uint32).astype(np.uint8)
    dev = ctx.alloc(n)
    assert ctx.lib.xvcgpu_memcpy_h2d_async(ctx.h, dev.ptr, hp, n) == 0
    ctx.sync()
    assert np.array_equal(dev.to_array(np.uint8, n), host)
    assert ctx.lib.xvcgpu_host_alloc(ctx.h, 0, C.byref(hp)) != 0
    dev.free()
    del host
    assert ctx.lib.xvcgpu_host_free(ctx.h, hp) == 0


def test_intra_recon_waves_arguments(gpu):
    api, ctx = gpu
    a, b = ctx.picture(64, 64, 10), ctx.picture(64, 64, 8)
    f = ctx.lib.xvcgpu_intra_recon_waves
    assert f(ctx.h, a.h_pic, a.h_pic, None, None, None, 0, None, None, None) == 0      # nothing to do
    assert f(ctx.h, a.h_pic, a.h_pic, None, None, None, 3, None, None, None) != 0      # lists missing
    d = ctx.alloc(256)
    assert f(ctx.h, a.h_pic, b.h_pic, d.ptr, d.ptr, d.ptr, 1, d.ptr, d.ptr, d.ptr) != 0  # bit depths
    d.free()
    a.destroy()
    b.destroy()
