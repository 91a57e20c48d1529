"""CTU-row sharding of the frame pass (xvc_amd/sharded.py): the orchestration
(shard plan, halo exchange, redundant boundary edge, gather) is run with a CPU
engine built on the oracle and must reproduce the unsharded frame pass bit for
bit - in one process with a loop-back comm (3 shards) and as two real
processes over gloo."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import oracle_frame
import oracle_lib as ol
from xvc_amd import api, pipeline, sharded, synth

BL, BC = 128, 64
PW, PH, BD, QP = 208, 112, 10, 32


def pad_planes(planes):
    return [np.ascontiguousarray(np.pad(p, BL if c == 0 else BC, mode="edge"))
            for c, p in enumerate(planes)]


class OracleEngine:
    """Same interface as sharded.GpuEngine, computed by the CPU oracle on numpy
    planes (shared with torch CPU tensors for the exchange)."""

    def __init__(self, lib, width, height, bd, qp, row_range, cu=16, search_range=96):
        import torch
        self.torch = torch
        self.lib, self.w, self.h, self.bd, self.cu = lib, width, height, bd, cu
        self.desc = pipeline.FrameDescriptors(width, height, qp, cu, search_range,
                                              row_range=row_range)
        self.cus_per_row = self.desc.cus_per_row
        self.pics = [[np.zeros(((height >> (c > 0)) + 2 * (BL >> (c > 0)),
                                (width >> (c > 0)) + 2 * (BL >> (c > 0))), np.uint16)
                      for c in range(3)] for _ in range(2)]
        self.cus = np.zeros(self.desc.n_cus_total, ol.CU_DTYPE)
        self.ssd_out = None
        self.ssd_t = torch.zeros(2, dtype=torch.int64)
        self._parts = pipeline.cu_partition(width, height, cu)

    def min_cu_height_at(self, y):
        if y <= 0 or y >= self.h:
            return 64
        hs = [p[3] for p in self._parts if p[1] == y or p[1] + p[3] == y]
        return min(hs) if hs else 64

    def encode(self, orig, ref_idx, rec_idx, ref_poc):
        # ME + MC + residual + metadata for the own CUs via the oracle's frame
        # pass restricted to the shard, without deblock/pad (done by phases)
        rec, res, nnz, cus, _ = oracle_frame.frame_pass(
            self.desc, self.bd, orig, self.pics[ref_idx], BL, ref_poc, lib=self.lib,
            encode_only=True)
        d = self.desc
        y0, y1 = d.row_range
        for c in range(3):
            b, s = (BL, 0) if c == 0 else (BC, 1)
            self.pics[rec_idx][c][b + (y0 >> s):b + (y1 >> s), :] = \
                rec[c][b + (y0 >> s):b + (y1 >> s), :]
        self.cus[d.cu_base:d.cu_base + d.n_cus] = cus[d.cu_base:d.cu_base + d.n_cus]

    def _planes(self, idx):
        pp = (ol.u16p * 3)()
        ss = (ol.pd * 3)()
        for c in range(3):
            b = BL if c == 0 else BC
            a = self.pics[idx][c]
            pp[c] = C.cast(a.ctypes.data + (b * a.strides[0] + b * 2), ol.u16p)
            ss[c] = a.strides[0] // 2
        return pp, ss

    def deblock_rows(self, rec_idx, pass_, ya, yb):
        f = self.lib.dll.xo_deblock_rows
        f.restype = None
        f.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(ol.u16p), C.POINTER(ol.pd),
                                      C.c_int, C.c_int, C.c_int]
        pp, ss = self._planes(rec_idx)
        cm = np.ascontiguousarray(self.desc.cu_map, np.int32)
        f(self.bd, self.w, self.h, 0, 0, 0, 4, self.cus.ctypes.data, cm.ctypes.data,
          cm.shape[1], pp, ss, pass_, ya, yb)

    def pad(self, rec_idx):
        self.lib.pad_border(self.w, self.h, self.pics[rec_idx], [BL, BC, BC])

    def search_reach(self):
        return sharded.search_reach(self.desc)

    def ssd(self, orig, rec_idx, ya=0, yb=1 << 30):
        o = np.ascontiguousarray(orig[0][BL:BL + self.h, BL:BL + self.w])
        r = np.ascontiguousarray(self.pics[rec_idx][0][BL:BL + self.h, BL:BL + self.w])
        self.ssd_out = self.lib.picture_ssd(self.bd, o, r, ya, yb)
        self.ssd_t[0], self.ssd_t[1] = self.ssd_out

    def ssd_tensor(self):
        return self.ssd_t

    def row_slab(self, rec_idx, comp, ya, yb):
        b = BL if comp == 0 else BC
        a = self.pics[rec_idx][comp]
        return self.torch.from_numpy(a[b + ya:b + yb, :].view(np.uint8)).reshape(-1)

    def cu_slab(self, first_cu, n):
        return self.torch.from_numpy(self.cus[first_cu:first_cu + n].view(np.uint8)) \
            .reshape(-1)


class LoopbackComm:
    """All ranks in one process: collect every rank's ops, then copy."""

    def __init__(self):
        self.pending = {}

    def exchange(self, sends, recvs):
        raise RuntimeError("use exchange_all")

    @staticmethod
    def exchange_all(ops_by_rank):
        # queue per (src, dst) in issue order
        queues = {}
        for src, (sends, _) in ops_by_rank.items():
            for dst, t in sends:
                queues.setdefault((src, dst), []).append(t.clone())
        for dst, (_, recvs) in ops_by_rank.items():
            for src, t in recvs:
                t.copy_(queues[(src, dst)].pop(0))
        assert all(len(q) == 0 for q in queues.values())


def reference_frames(lib, n_frames, w=PW, h=PH, search_range=96):
    """Unsharded oracle chain: returns list of (rec planes, ssd)."""
    clip = synth.SyntheticClip(w, h, BD)
    desc = pipeline.FrameDescriptors(w, h, QP, search_range=search_range)
    ref = pad_planes(clip.frame(0))
    out = []
    for n in range(1, n_frames + 1):
        orig = pad_planes(clip.frame(n))
        rec, _, _, _, ssd = oracle_frame.frame_pass(desc, BD, orig, ref, BL, n - 1, lib=lib)
        out.append((rec, ssd))
        ref = rec
    return out


def assert_valid_rows_equal(s, planes, expect, h, tag):
    """A shard holds the reconstruction only as far as its next search can
    reach (ShardedFramePass.valid_rows); those rows - and the top / bottom
    border when they are picture edges - must match the unsharded result."""
    ya, yb = s.valid_rows()
    for c in range(3):
        b, sh = (BL, 0) if c == 0 else (BC, 1)
        lo = 0 if ya == 0 else b + (ya >> sh)
        hi = planes[c].shape[0] if yb == h else b + (yb >> sh)
        assert np.array_equal(planes[c][lo:hi], expect[c][lo:hi]), (tag, s.rank, c)


def test_shard_rows():
    assert sharded.shard_rows(1080, 8) == [(0, 144), (144, 288), (288, 432), (432, 576),
                                           (576, 704), (704, 832), (832, 960), (960, 1080)]
    assert sharded.shard_rows(1080, 1) == [(0, 1080)]
    for h, w in ((1080, 8), (2160, 8), (288, 2), (112, 3), (4320, 8)):
        r = sharded.shard_rows(h, w)
        assert r[0][0] == 0 and r[-1][1] == h
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(a[0] % 16 == 0 for a in r)


@pytest.mark.parametrize("w,h,search_range,world", [
    (PW, PH, 96, 2), (PW, PH, 96, 3),
    # tall picture, short search: only neighbouring shards exchange rows
    (208, 320, 8, 4)])
def test_sharded_loopback_matches_unsharded(w, h, search_range, world):
    lib = ol.Lib("xo")
    expect = reference_frames(lib, 3, w, h, search_range)
    clip = synth.SyntheticClip(w, h, BD)
    rows = sharded.shard_rows(h, world)
    ranks = []
    for r in range(world):
        e = OracleEngine(lib, w, h, BD, QP, rows[r], search_range=search_range)
        for c, p in enumerate(pad_planes(clip.frame(0))):
            e.pics[0][c][:] = p
        ranks.append(sharded.ShardedFramePass(e, LoopbackComm(), r, world))
    if h == 320:
        # the point of the case: no traffic between non-adjacent shards
        for s in ranks:
            sends, recvs = s.gather_ops(0)
            assert {p for p, _ in sends} | {p for p, _ in recvs} <= {s.rank - 1, s.rank + 1}
            assert s.valid_rows() != (0, h)
    for n in (1, 2, 3):
        orig = pad_planes(clip.frame(n))
        ref_idx, rec_idx = (n - 1) % 2, n % 2
        for s in ranks:
            s.phase_a(orig, ref_idx, rec_idx, n - 1)
        LoopbackComm.exchange_all({s.rank: s.halo_ops(rec_idx) for s in ranks})
        for s in ranks:
            s.phase_b(rec_idx)
        LoopbackComm.exchange_all({s.rank: s.gather_ops(rec_idx) for s in ranks})
        for s in ranks:
            s.phase_c(orig, rec_idx)
        exp_rec, exp_ssd = expect[n - 1]
        for s in ranks:
            assert_valid_rows_equal(s, s.e.pics[rec_idx], exp_rec, h, (world, n))
        # the per-shard PSNR parts add up to the picture's
        assert tuple(sum(s.e.ssd_out[k] for s in ranks) for k in (0, 1)) == exp_ssd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q, w=PW, h=PH, search_range=96):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ol.Lib("xo")
    clip = synth.SyntheticClip(w, h, BD)
    rows = sharded.shard_rows(h, world)
    e = OracleEngine(lib, w, h, BD, QP, rows[rank], search_range=search_range)
    for c, p in enumerate(pad_planes(clip.frame(0))):
        e.pics[0][c][:] = p
    # packed: one all_to_all_single per exchange (what runs over RCCL)
    s = sharded.ShardedFramePass(e, sharded.TorchComm(dist, rank, world), rank, world)
    # a second, independent chain with its own process group, interleaved with
    # the first (bench.py --chains): same pictures, so the same results; this
    # one with a point-to-point operation per slab
    e2 = OracleEngine(lib, w, h, BD, QP, rows[rank], search_range=search_range)
    for c, p in enumerate(pad_planes(clip.frame(0))):
        e2.pics[0][c][:] = p
    s2 = sharded.ShardedFramePass(e2, sharded.TorchComm(dist, rank, world, dist.new_group(),
                                                        packed=False), rank, world)
    out = []
    for n in (1, 2):
        orig = pad_planes(clip.frame(n))
        s.run(orig, (n - 1) % 2, n % 2, n - 1)
        s2.run(orig, (n - 1) % 2, n % 2, n - 1)
        out.append(([p.copy() for p in e.pics[n % 2]], s.total_ssd(), s.valid_rows()))
        assert all(np.array_equal(a, b) for a, b in zip(e.pics[n % 2], e2.pics[n % 2]))
        assert s2.total_ssd() == out[-1][1]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gloo_two_ranks():
    import torch.multiprocessing as mp
    lib = ol.Lib("xo")
    expect = reference_frames(lib, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        for n in range(2):
            rec, ssd, valid = got[rank][n]
            assert valid == (0, PH)   # search range 96 spans this small picture
            for c in range(3):
                assert np.array_equal(rec[c], expect[n][0][c]), (rank, n, c)
            assert ssd == expect[n][1]   # all-reduced over the two ranks


def test_sharded_gloo_four_ranks_neighbours_only():
    """Tall picture, short search: a rank exchanges rows with its neighbours
    only, so the all-to-all has empty segments for everybody else."""
    import torch.multiprocessing as mp
    w, h, sr, world = 208, 320, 8, 4
    lib = ol.Lib("xo")
    expect = reference_frames(lib, 2, w, h, sr)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, w, h, sr))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert any(got[r][0][2] != (0, h) for r in range(world))
    for rank in range(world):
        for n in range(2):
            rec, ssd, (ya, yb) = got[rank][n]
            for c in range(3):
                b = BL if c == 0 else BC   # plane border rows
                lo, hi = (ya, yb) if c == 0 else (ya // 2, yb // 2)
                assert np.array_equal(rec[c][b + lo:b + hi], expect[n][0][c][b + lo:b + hi]), \
                    (rank, n, c)
            assert ssd == expect[n][1]


def test_synthetic_clip_c_mirror_matches_python():
    """the clip generator exists in C (oracle/xvc_synth.c) and in
    Python (xvc_amd/synth.py) and both produce identical bytes."""
    import ctypes as C
    import oracle_lib as ol
    from xvc_amd import synth
    dll = ol.Lib("xo").dll
    dll.xo_synth_frame.restype = None
    dll.xo_synth_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                   ol.u16p, ol.pd, ol.u16p, ol.pd, ol.u16p, ol.pd]
    for (w, h, bd, square) in [(136, 72, 10, True), (352, 288, 8, True), (64, 48, 12, False)]:
        clip = synth.SyntheticClip(w, h, bd, square=square)
        for n in (0, 1, 7, 33, 70):
            exp = clip.frame(n)
            y = np.zeros((h, w), np.uint16)
            u = np.zeros((h // 2, w // 2), np.uint16)
            v = np.zeros((h // 2, w // 2), np.uint16)
            dll.xo_synth_frame(w, h, bd, 1234, int(square), n, ol.ptr(y, ol.u16p), w,
                               ol.ptr(u, ol.u16p), w // 2, ol.ptr(v, ol.u16p), w // 2)
            assert np.array_equal(y, exp[0]) and np.array_equal(u, exp[1]) and \
                np.array_equal(v, exp[2]), (w, h, bd, n)
