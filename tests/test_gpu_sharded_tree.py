"""SURVEY 8e scheme (A) with the HIP engine: the in-loop filter of real CU trees
sharded by CTU rows (sharded.ShardedTreeFilter + GpuTreeEngine =
xvcgpu_deblock_rows on row ranges), every shard a separate engine with its own
picture on this one GPU, rows exchanged by device copies instead of RCCL - CIF in
2 / 4 shards, the 1080p B pictures in 8 - and the C++ entry point of the product
path (xvc_host_shard_filter_run) as a one-rank job."""
import ctypes as C

import numpy as np
import pytest

import stream_fixture as sf
import test_sharded_tree as tst
from test_sharded import LoopbackComm
from xvc_amd import decoder, sharded

pytestmark = pytest.mark.gpu
BL = 128


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def pad(planes):
    return [np.ascontiguousarray(np.pad(p, BL >> (1 if c else 0), mode="edge"))
            for c, p in enumerate(planes)]


def run_shards(ctx, info, pre, post, cus, cu_map, rows, tag):
    import torch
    dev = torch.device("cuda", 0)
    w, h, bd = int(info["width"]), int(info["height"]), int(info["bitdepth"])
    args = (int(info["pic_type"]) == 0, int(info["beta_offset"]), int(info["tc_offset"]))
    world = len(rows) - 1
    ranks = []
    for r in range(world):
        e = sharded.GpuTreeEngine(ctx, w, h, bd, cus, cu_map, *args, dev)
        e.picture.upload(pad(tst.own_rows_only(pre, rows[r], rows[r + 1], 31 * r + 5)), BL)
        ranks.append(sharded.ShardedTreeFilter(e, LoopbackComm(), r, world, rows))
    for s in ranks:
        s.step_local()
    ctx.sync()
    LoopbackComm.exchange_all({s.rank: s.ops_down() for s in ranks})
    torch.cuda.synchronize()
    for s in ranks:
        s.step_strip()
    ctx.sync()
    LoopbackComm.exchange_all({s.rank: s.ops_up() for s in ranks})
    torch.cuda.synchronize()
    chains = 0
    for s in ranks:
        got = s.e.picture.download(0)
        for c in range(3):
            a, b = (s.y0, s.y1) if c == 0 else (s.y0 // 2, s.y1 // 2)
            assert np.array_equal(got[c][a:b], post[c][a:b]), (tag, s.rank, c)
        chains += s.d_top > 4          # 4 = the boundary edge alone
        s.e.destroy()
    return chains


@pytest.mark.parametrize("world", [2, 4])
def test_ordered_handoff_cif(gpu, world):
    api, ctx = gpu
    chains = 0
    for k, (info, pre, post, cus, cu_map) in enumerate(tst.b_pictures("c0")):
        rows = tst.boundaries(int(info["height"]), world)
        chains += run_shards(ctx, info, pre, post, cus, cu_map, rows, ("c0", k))
    assert chains >= 3


def test_cut_clears_every_band_gpu(gpu):
    """The synthetic tree of test_sharded_tree.test_cut_clears_every_band on the HIP
    engine: a band whose run ends early and starts again across the longest run's
    end - the sharded filter equals xvcgpu_deblock of the whole picture."""
    from helpers import make_cus
    import oracle_lib as ol
    import torch
    api, ctx = gpu
    pw, ph, y0, bd = 64, 128, 64, 10
    parts = tst._column_tree([[4, 4, 8, 16], [8, 4, 4, 16]], 32, pw, ph, y0)
    rows = [0, y0, ph]
    for trial in range(6):
        rng = np.random.default_rng(900 + trial)
        cus, cu_map = make_cus(rng, parts, 0, [0], [8], pw, ph)
        cus["intra"], cus["qp_y"], cus["qp_c"] = 1, 40, ol.chroma_qp(40)
        pre = []
        for c in range(3):
            w, h = (pw, ph) if c == 0 else (pw // 2, ph // 2)
            base = rng.integers(0, 1 << bd, size=((h + 3) // 4, (w + 3) // 4))
            p = np.kron(base, np.ones((4, 4), np.int64))[:h, :w] // 8 + 400
            pre.append(np.clip(p + rng.integers(-3, 4, size=(h, w)), 0, 1023).astype(np.uint16))
        whole = sharded.GpuTreeEngine(ctx, pw, ph, bd, cus, cu_map, 0, 0, 0,
                                      torch.device("cuda", 0))
        whole.picture.upload(pad(pre), BL)
        whole.deblock_rows(0, 0, ph)
        whole.deblock_rows(1, 0, ph)
        ctx.sync()
        post = whole.picture.download(0)
        whole.destroy()
        assert not np.array_equal(post[0], pre[0])
        info = {"width": pw, "height": ph, "bitdepth": bd, "pic_type": 1, "beta_offset": 0,
                "tc_offset": 0}
        assert run_shards(ctx, info, pre, post, cus, cu_map, rows, ("synthetic", trial)) == 1


def _decode_variants(ctx, name):
    """Per single-tree picture of the stream: (info, unfiltered planes, final planes)
    from the device decoder (final: MD5-checked against the stream)."""
    fx = sf.StreamFixture(name)
    w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
    dec = decoder.PictureDecoder(ctx, w, h, bd)
    done, out = {}, []
    scratch = ctx.picture(w, h, bd)
    for i in range(fx.n):
        info = fx.info[i]
        ps, cs = sf.to_syntax(info, fx.cus(i))
        refs = [[done[int(info["ref_poc"][l][k])] for k in range(int(info["num_ref"][l]))]
                for l in range(2)]
        rec = ctx.picture(w, h, bd)
        dec.decode(ps, cs, fx.levels(i), refs, rec)
        ctx.sync()
        post = rec.download(0)
        assert np.array_equal(sf.picture_md5(post, bd), info["md5"])
        done[int(info["poc"])] = rec
        if int(info["two_trees"]) or not int(info["deblock"]):
            continue
        ps2 = ps.copy()
        ps2["deblock"] = 0
        dec.decode(ps2, cs, fx.levels(i), refs, scratch)
        ctx.sync()
        out.append((info, scratch.download(0), post, *sf.deblock_metadata(info, fx.cus(i))))
    scratch.destroy()
    dec.destroy()
    return out, done


def test_ordered_handoff_1080p_eight_shards(gpu):
    """BASELINE config 1's real B pictures in 8 CTU-row shards (135 rows do not
    divide into CTU rows: 2 rows of CTUs per rank, the last rank takes the rest)."""
    api, ctx = gpu
    pics, done = _decode_variants(ctx, "c1")
    assert len(pics) == 4
    for k, (info, pre, post, cus, cu_map) in enumerate(pics):
        h = int(info["height"])
        rows = [128 * r for r in range(8)] + [h]
        run_shards(ctx, info, pre, post, cus, cu_map, rows, ("c1", k))
    for p in done.values():
        p.destroy()


def test_cpp_shard_filter_one_rank(gpu):
    """xvc_host_shard_filter_run (the RCCL product path) as a one-rank job = the
    whole picture's filter."""
    api, ctx = gpu
    L = decoder.load_host_library()
    L.xvc_host_shard_filter_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + \
        [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 4
    for k, (info, pre, post, cus, cu_map) in enumerate(tst.b_pictures("c0")[:3]):
        w, h, bd = int(info["width"]), int(info["height"]), int(info["bitdepth"])
        pic = ctx.picture(w, h, bd)
        pic.upload(pad(pre), BL)
        d_cus, d_map = ctx.buffer(cus), ctx.buffer(cu_map)
        rows = np.array([0, h], np.int32)
        m = np.ascontiguousarray(cu_map, np.int32)
        st = L.xvc_host_shard_filter_run(ctx.h, None, 0, 1, rows.ctypes.data, pic.h_pic,
                                         d_cus.ptr, len(cus), d_map.ptr, m.ctypes.data,
                                         m.shape[1], int(info["pic_type"]) == 0,
                                         int(info["beta_offset"]), int(info["tc_offset"]))
        assert st == 0
        ctx.sync()
        got = pic.download(0)
        for c in range(3):
            assert np.array_equal(got[c], post[c]), (k, c)
        for b in (d_cus, d_map):
            b.free()
        pic.destroy()
