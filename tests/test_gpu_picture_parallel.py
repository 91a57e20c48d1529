"""Picture-level parallel frame passes on the device: the GPU engine of
xvc_amd/picture_parallel.py (picture slots = streams, ring of reconstructions,
events) against the oracle engine of tests/test_picture_parallel.py, and the
RCCL exchange of libxvcgpu.so with the one rank a single GPU allows (a rank
sending to itself inside a group: the same ncclSend / ncclRecv path)."""
import numpy as np
import pytest

import test_picture_parallel as tpp
from xvc_amd import picture_parallel, schedule, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


@pytest.mark.parametrize("rdoq", [False, True])
@pytest.mark.parametrize("slots", [1, 3])
def test_gpu_engine_matches_oracle_engine(gpu, slots, rdoq):
    api, ctx = gpu
    s1 = schedule.Schedule(tpp.N_PICTURES, tpp.SUB_GOP, 2, 1, 1)
    o = tpp.OraclePictureEngine(s1, 0, rdoq=rdoq)
    picture_parallel.run_rank(s1, 0, o)
    base = {int(s1.pictures[i]["poc"]): pl for i, pl in o.coded.items()}

    s = schedule.Schedule(tpp.N_PICTURES, tpp.SUB_GOP, 2, 1, slots)
    clip = synth.SyntheticClip(tpp.PW, tpp.PH, tpp.BD)
    origs = []
    for poc in range(tpp.N_PICTURES):
        pic = ctx.picture(tpp.PW, tpp.PH, tpp.BD)
        pic.upload(tpp.pad_planes(clip.frame(poc)), tpp.BL)
        origs.append(pic)

    class AllOrigs(picture_parallel.GpuPictureEngine):
        def _orig(self, poc):
            return self.origs[poc]
    e = AllOrigs(ctx, s, 0, tpp.PW, tpp.PH, tpp.BD, tpp.QP, origs, rdoq=rdoq)
    # keep every picture: download right after its entry was written (the ring
    # is smaller than the sequence when several sub-GOPs are in flight)
    got = {}

    def keep(index):
        got[int(s.pictures[index]["poc"])] = e.download(index)
    e.after_encode = keep
    picture_parallel.run_rank(s, 0, e)
    assert len(got) == tpp.N_PICTURES
    for poc, planes in got.items():
        for c in range(3):
            assert np.array_equal(planes[c], base[poc][c]), (poc, c)
    for pic in origs:
        pic.destroy()


def test_gpu_engine_overlaps_without_downloads(gpu):
    """The same walk left asynchronous (no download between pictures): slots
    overlap, ring entries are reused; the last sub-GOP's pictures still match."""
    api, ctx = gpu
    n = 1 + 8 * 12
    s = schedule.Schedule(n, 8, 2, 1, 3)
    clip = synth.SyntheticClip(tpp.PW, tpp.PH, tpp.BD)
    origs = []
    for k in range(6):
        pic = ctx.picture(tpp.PW, tpp.PH, tpp.BD)
        pic.upload(tpp.pad_planes(clip.frame(k)), tpp.BL)
        origs.append(pic)
    e = picture_parallel.GpuPictureEngine(ctx, s, 0, tpp.PW, tpp.PH, tpp.BD, tpp.QP, origs)
    assert e.ring < n
    picture_parallel.run_rank(s, 0, e)
    e.sync()
    # oracle with the same cyclic originals
    s1 = schedule.Schedule(n, 8, 2, 1, 1)
    o = tpp.OraclePictureEngine(s1, 0, rdoq=True)
    frames = [clip.frame(k) for k in range(6)]

    class Cyc:
        def frame(self, poc):
            k = poc % 10
            return frames[k if k < 6 else 10 - k]
    o.clip = Cyc()
    picture_parallel.run_rank(s1, 0, o)
    checked = 0
    for i, p in enumerate(s.pictures):
        if e.holds[i % e.ring] == i:
            planes = e.download(i)
            j = s1.index_of_poc[int(p["poc"])]
            for c in range(3):
                assert np.array_equal(planes[c], o.coded[j][c]), (int(p["poc"]), c)
            checked += 1
    assert checked >= 8


def test_rccl_self_exchange(gpu):
    """One rank, the native RCCL path: a picture and a band of rows sent to
    itself inside a group arrive intact, ordered by events."""
    api, ctx = gpu
    comm = api.Comm(ctx, api.comm_unique_id(), 1, 0)
    assert ctx.lib.xvcgpu_comm_world(comm.h) == 1 and ctx.lib.xvcgpu_comm_rank(comm.h) == 0
    rng = np.random.default_rng(5)
    w, h, bd = 208, 112, 10
    planes = [rng.integers(0, 1 << bd, ((h >> (c > 0)) + 2 * (128 >> (c > 0)),
                                         (w >> (c > 0)) + 2 * (128 >> (c > 0)))).astype(np.uint16)
              for c in range(3)]
    A, B, Cc = ctx.picture(w, h, bd), ctx.picture(w, h, bd), ctx.picture(w, h, bd)
    A.upload(planes, 128)
    zeros = [np.zeros_like(p) for p in planes]
    B.upload(zeros, 128)
    Cc.upload(zeros, 128)
    ready, done = api.Event(ctx), api.Event(ctx)
    ready.record(ctx)
    comm.wait_event(ready)
    comm.group_begin()
    comm.send_picture(A, 0)
    comm.recv_picture(B, 0)
    comm.group_end()
    comm.group_begin()
    comm.send_rows(A, 16, 24, 0)
    comm.recv_rows(Cc, 16, 24, 0)
    comm.group_end()
    comm.record_event(done)
    done.wait(ctx)
    got = B.download(128)
    for c in range(3):
        assert np.array_equal(got[c], planes[c])
    rows = Cc.download(128)
    for c in range(3):
        b, sh = (128, 0) if c == 0 else (64, 1)
        exp = np.zeros_like(planes[c])
        exp[b + (16 >> sh):b + (24 >> sh), :] = planes[c][b + (16 >> sh):b + (24 >> sh), :]
        assert np.array_equal(rows[c], exp), c
    # counters
    buf = ctx.buffer(np.array([5, 7], np.uint64))
    comm.all_reduce_sum_u64(buf.ptr, 2)
    comm.sync()
    assert buf.to_array(np.uint64, 2).tolist() == [5, 7]
    buf.free()
    for x in (ready, done):
        x.destroy()
    comm.destroy()
    for x in (A, B, Cc):
        x.destroy()


def test_bench_picture_mode_same_pictures_for_any_world(tmp_path):
    """bench.py's picture-level mode as real processes: 1 rank, and 2 ranks sharing
    this GPU (pictures staged through the host - RCCL refuses two ranks on one
    device): the checksums of the last pictures agree."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["bench.py", "--schedule", "subgop", "--no-cpu", "--width", "352", "--height",
              "288", "--steps", "40", "--warmup", "17", "--frames", "5"]
    one = subprocess.run([sys.executable] + common, cwd=root, capture_output=True, text=True)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    env = dict(os.environ, XVC_BENCH_BACKEND="gloo", XVC_BENCH_DEVICE="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541"] + common + ["--gpus", "2"],
                         cwd=root, capture_output=True, text=True, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert a["tail_crc"] == b["tail_crc"] and len(a["tail_crc"]) == 8
    assert a["rccl_world_size"] == 1 and b["n_gpus"] == 2


def test_native_row_comm_self_exchange(gpu):
    """sharded.NativeComm (the row shards' exchange on libxvcgpu.so's
    communicator): slabs sent to the own rank arrive, ordered after the
    producing kernel and before the consuming copy."""
    import torch
    from xvc_amd import sharded
    api, ctx = gpu
    comm = api.Comm(ctx, api.comm_unique_id(), 1, 0)
    nc = sharded.NativeComm(ctx, comm)
    a = torch.arange(0, 1 << 16, dtype=torch.int32, device="cuda").view(torch.uint8)
    b = torch.zeros_like(a)
    c2 = torch.full((64,), 7, dtype=torch.uint8, device="cuda")
    d = torch.zeros_like(c2)
    torch.cuda.synchronize()
    nc.exchange([(0, a), (0, c2)], [(0, b), (0, d)])
    ctx.sync()
    assert torch.equal(a, b) and torch.equal(c2, d)
    t = torch.tensor([3, 9], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert nc.allreduce_sum(t).tolist() == [3, 9]
    comm.destroy()
