"""One step of a CU state's SearchMotion into all its reference pictures as ONE launch
(xvcgpu_me_search_refs / _bipred_search_refs / _mc_metric_batch_refs /
_affine_me_batch_refs: the pictures as a table, a slot byte per job, only the CU's
block class launched) against the single-picture entry points job by job - which the
other tests hold against the oracle and the reference's captured calls - and, for the
uni-directional search, directly against the oracle."""
import numpy as np
import pytest

from xvc_amd import synth

pytestmark = pytest.mark.gpu

W, H, BD, BL = 352, 288, 10, 128
CLASS_SIZES = {16: [(16, 16), (8, 8), (16, 8), (8, 16), (16, 4), (4, 8)],
               32: [(32, 32), (32, 16), (16, 32), (32, 8)],
               64: [(64, 64), (64, 32), (32, 64), (64, 16)]}


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    clip = synth.SyntheticClip(W, H, BD)
    pad = lambda planes: [np.ascontiguousarray(np.pad(p, BL >> (c > 0), mode="edge"))
                          for c, p in enumerate(planes)]
    pics = []
    for k in range(4):
        p = ctx.picture(W, H, BD)
        p.host_luma = pad(clip.frame(k))[0]       # (for the oracle)
        p.upload(pad(clip.frame(k)), BL)
        pics.append(p)
    yield api, ctx, pics[3], pics[:3]
    for p in pics:
        p.destroy()
    ctx.close()


def _me_blocks(api, rng, cls, n):
    b = np.zeros(n, api.ME_DTYPE)
    for i in range(n):
        w, h = CLASS_SIZES[cls][int(rng.integers(len(CLASS_SIZES[cls])))]
        b[i]["w"], b[i]["h"] = w, h
        b[i]["x"] = int(rng.integers(0, (W - w) // 8 + 1)) * 8
        b[i]["y"] = int(rng.integers(0, (H - h) // 8 + 1)) * 8
    b["depth_nonzero"] = rng.integers(0, 2, n)
    b["fullpel_mv"] = rng.integers(0, 2, n)
    b["mvp_x"], b["mvp_y"] = rng.integers(-96, 97, n), rng.integers(-96, 97, n)
    b["prev_x"], b["prev_y"] = rng.integers(-3, 4, n), rng.integers(-3, 4, n)
    b["lambda16"] = rng.choice([120000, 498000, 1500000], n)
    b["search_range"] = rng.choice([96, 128], n)
    return b


def _slots(rng, n, width=1):
    s = rng.integers(0, 3, (n, width)).astype(np.uint8)
    s[rng.random(n) < 0.15, 0] = 255          # no job
    return s


@pytest.mark.parametrize("cls", [16, 32, 64])
def test_me_search_refs(gpu, cls):
    api, ctx, orig, refs = gpu
    rng = np.random.default_rng(100 + cls)
    n = 40
    blocks, slots = _me_blocks(api, rng, cls, n), _slots(rng, n)[:, 0]
    before = np.zeros(n, api.MERES_DTYPE)
    before["fullpel_cost"] = 0x5a5a5a5a
    got = ctx.me_search_refs(orig, refs, blocks, slots, cls, results=before)
    for s in range(3):
        idx = np.flatnonzero(slots == s)
        exp = ctx.me_search(orig, refs[s], blocks[idx])
        assert np.array_equal(got[idx], exp), (cls, s)
        assert (exp["fullpel_cost"] != 0xffffffff).all()
    none = slots == 255
    assert none.any() and np.array_equal(got[none], before[none])
    # and directly against the oracle (TzSearch::Search + SubpelSearch on the job's picture)
    import oracle_lib as ol
    from test_gpu_parity import to_me_struct
    xo = ol.Lib("xo")
    for i in np.flatnonzero(slots != 255):
        st = to_me_struct(blocks[i])
        rp = refs[int(slots[i])].host_luma
        (fx, fy), cost = xo.tz_search(BD, st, W, H, orig.host_luma, rp, BL)
        assert (int(got[i]["fullpel_x"]), int(got[i]["fullpel_y"]), int(got[i]["fullpel_cost"])) == (fx, fy, cost), \
            (cls, i, tuple(blocks[i]), tuple(got[i]))
        if not (blocks[i]["fullpel_mv"] & 1):
            (sx, sy), sd = xo.subpel_search(BD, st, W, H, orig.host_luma, rp, BL, (fx, fy))
            assert (int(got[i]["mv_x"]), int(got[i]["mv_y"]), int(got[i]["subpel_dist"])) == (sx, sy, sd), \
                (cls, i, tuple(blocks[i]), tuple(got[i]))


@pytest.mark.parametrize("cls", [16, 32, 64])
def test_bipred_search_refs(gpu, cls):
    api, ctx, orig, refs = gpu
    rng = np.random.default_rng(200 + cls)
    n = 36
    jobs = np.zeros(n, api.BI_DTYPE)
    jobs["blk"] = _me_blocks(api, rng, cls, n)
    jobs["blk"]["search_range"] = 4
    for f in ("other_mv_x", "other_mv_y", "boot_mv_x", "boot_mv_y"):
        jobs[f] = rng.integers(-200, 201, n)
    slots = _slots(rng, n, 2)
    before = np.zeros(n, api.MERES_DTYPE)
    before["subpel_dist"] = 0x5a5a5a5a
    got = ctx.bipred_search_refs(orig, refs, jobs, slots, cls, results=before)
    for s in range(3):
        for o in range(3):
            idx = np.flatnonzero((slots[:, 0] == s) & (slots[:, 1] == o))
            if not len(idx):
                continue
            exp = ctx.bipred_search(orig, refs[o], refs[s], jobs[idx])
            assert np.array_equal(got[idx], exp), (cls, s, o)
    none = slots[:, 0] == 255
    assert none.any() and np.array_equal(got[none], before[none])


def test_mc_metric_batch_refs(gpu):
    api, ctx, orig, refs = gpu
    rng = np.random.default_rng(300)
    n = 64
    c = np.zeros(n, api.MCM_DTYPE)
    sizes = [s for v in CLASS_SIZES.values() for s in v]
    for i in range(n):
        w, h = sizes[int(rng.integers(len(sizes)))]
        c[i]["w"], c[i]["h"] = w, h
        c[i]["x"] = int(rng.integers(0, (W - w) // 8 + 1)) * 8
        c[i]["y"] = int(rng.integers(0, (H - h) // 8 + 1)) * 8
    c["metric"] = rng.choice([0, 3, 5], n)
    c["qp"] = 32
    c["mv_x"], c["mv_y"] = rng.integers(-300, 301, n), rng.integers(-300, 301, n)
    slots = _slots(rng, n)[:, 0]
    got = ctx.mc_metric_batch_refs(orig, refs, c, slots)
    for s in range(3):
        idx = np.flatnonzero(slots == s)
        assert np.array_equal(got[idx], ctx.mc_metric_batch(orig, refs[s], c[idx])), s
    assert (got[slots == 255] == 0xffffffffffffffff).all()


@pytest.mark.parametrize("cu_height", [16, 32, 64])
def test_affine_me_batch_refs(gpu, cu_height):
    api, ctx, orig, refs = gpu
    rng = np.random.default_rng(400 + cu_height)
    n = 18
    b = np.zeros(n, api.AFFINE_ME_DTYPE)
    b["h"] = cu_height
    b["w"] = rng.choice([16, 32, 64], n)
    for i in range(n):
        b[i]["x"] = int(rng.integers(0, (W - int(b[i]["w"])) // 8 + 1)) * 8
        b[i]["y"] = int(rng.integers(0, (H - cu_height) // 8 + 1)) * 8
    b["flags"] = rng.integers(0, 4, n)
    b["lambda16"] = 90000
    b["mvp"] = rng.integers(-64, 65, (n, 3, 2))
    b["bootstrap"] = b["mvp"] + rng.integers(-16, 17, (n, 3, 2))
    b["other_mv"] = rng.integers(-64, 65, (n, 3, 2))
    slots = _slots(rng, n, 2)
    got = ctx.affine_me_batch_refs(orig, refs, b, slots, cu_height)
    for s in range(3):
        for o in range(3):
            idx = np.flatnonzero((slots[:, 0] == s) & (slots[:, 1] == o))
            if not len(idx):
                continue
            exp = ctx.affine_me_batch(orig, refs[s], b[idx], ref_other=refs[o])
            assert np.array_equal(got[idx], exp), (cu_height, s, o)
            assert (exp["dist"] != 0xffffffff).all()
    none = slots[:, 0] == 255
    assert (got[none]["dist"] == 0).all()


def test_slot_beyond_the_table_is_no_job(gpu):
    """A slot >= n_refs (here 2 of a table of 2) is not searched, whatever lies behind
    the table's entries in use."""
    api, ctx, orig, refs = gpu
    rng = np.random.default_rng(7)
    blocks = _me_blocks(api, rng, 16, 12)
    slots = np.array([0, 1, 2, 9, 255, 0, 1, 2, 0, 1, 3, 0], np.uint8)
    before = np.zeros(12, api.MERES_DTYPE)
    before["fullpel_cost"] = 0x5a5a5a5a
    got = ctx.me_search_refs(orig, refs[:2], blocks, slots, 16, results=before)
    for s_ in range(2):
        idx = np.flatnonzero(slots == s_)
        assert np.array_equal(got[idx], ctx.me_search(orig, refs[s_], blocks[idx]))
    none = slots >= 2
    assert np.array_equal(got[none], before[none])


def test_refs_forms_refuse_bad_arguments(gpu):
    api, ctx, orig, refs = gpu
    arr = ctx._ref_array(refs)
    d = ctx.alloc(4096)
    L = ctx.lib
    assert L.xvcgpu_me_search_refs(ctx.h, orig.h_pic, arr, 3, 3, d.ptr, d.ptr, 1, d.ptr, 24) != 0
    assert L.xvcgpu_me_search_refs(ctx.h, orig.h_pic, arr, 11, 3, d.ptr, d.ptr, 1, d.ptr, 16) != 0
    assert L.xvcgpu_me_search_refs(ctx.h, orig.h_pic, arr, 3, 3 | api.ME_LIC_JOBS, d.ptr, d.ptr, 1,
                                   d.ptr, 16) != 0
    assert L.xvcgpu_bipred_search_refs(ctx.h, orig.h_pic, arr, 0, d.ptr, d.ptr, 1, d.ptr, 16) != 0
    assert L.xvcgpu_affine_me_batch_refs(ctx.h, orig.h_pic, arr, 3, d.ptr, d.ptr, 1, d.ptr, 8) != 0
    assert L.xvcgpu_mc_metric_batch_refs(ctx.h, orig.h_pic, None, 3, 16, d.ptr, d.ptr, 1, d.ptr) != 0
    d.free()
