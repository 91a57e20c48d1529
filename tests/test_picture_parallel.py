"""Picture-level parallel frame passes (xvc_amd/picture_parallel.py over the
C++ schedule): ranks code independent pictures of the sub-GOPs and ship the
reconstructed references.  Here with a CPU engine built on the oracle and
torch.distributed (gloo) as the transport, world sizes 2 and 4: every picture
any rank coded equals the picture of the single-rank walk, bit for bit; and
what a rank received equals what the owner coded."""
import os
import socket

import numpy as np
import pytest

import oracle_frame
import oracle_lib as ol
from xvc_amd import picture_parallel, pipeline, schedule, synth

BL, BC = 128, 64
PW, PH, BD, QP = 96, 64, 10, 32


def pad_planes(planes):
    return [np.ascontiguousarray(np.pad(p, BL if c == 0 else BC, mode="edge"))
            for c, p in enumerate(planes)]


class OraclePictureEngine:
    """encode / send / recv of picture_parallel.GpuPictureEngine on numpy planes,
    computed by the oracle's frame pass; transfers through `dist` (or none)."""

    def __init__(self, s, rank, dist=None, rdoq=False):
        self.s, self.rank, self.dist = s, rank, dist
        self.lib = ol.Lib("xo")
        self.desc = pipeline.FrameDescriptors(PW, PH, QP, rdoq=rdoq, bitdepth=BD)
        self.clip = synth.SyntheticClip(PW, PH, BD)
        self.ring = picture_parallel.ring_size(s)
        self.recs = [None] * self.ring
        self.holds = [-1] * self.ring
        self.coded = {}
        self.received = {}

    def encode(self, p, index, ref_indices):
        poc = int(p["poc"])
        orig = pad_planes(self.clip.frame(poc))
        if p["intra"]:
            rec = orig
        else:
            for j in ref_indices:       # every listed picture must be here, not only the one read
                assert self.holds[j % self.ring] == j, (index, j)
            r = ref_indices[0]
            rec = oracle_frame.frame_pass(self.desc, BD, orig, self.recs[r % self.ring], BL,
                                          int(self.s.pictures[r]["poc"]), lib=self.lib)[0]
        self.recs[index % self.ring] = rec
        self.holds[index % self.ring] = index
        self.coded[index] = [a.copy() for a in rec]

    def send(self, p, index, dst):
        import torch
        assert self.holds[index % self.ring] == index
        for a in self.recs[index % self.ring]:
            self.dist.send(torch.from_numpy(np.ascontiguousarray(a).view(np.int16)), dst)

    def recv(self, p, index, src):
        import torch
        rec = []
        for c in range(3):
            b = BL if c == 0 else BC
            a = np.zeros(((PH >> (c > 0)) + 2 * b, (PW >> (c > 0)) + 2 * b), np.uint16)
            self.dist.recv(torch.from_numpy(a.view(np.int16)), src)
            rec.append(a)
        self.recs[index % self.ring] = rec
        self.holds[index % self.ring] = index
        self.received[index] = [a.copy() for a in rec]


N_PICTURES, SUB_GOP = 1 + 8 * 3, 8


def single_rank_pictures():
    s = schedule.Schedule(N_PICTURES, SUB_GOP, 2, 1, 1)
    e = OraclePictureEngine(s, 0)
    picture_parallel.run_rank(s, 0, e)
    assert len(e.coded) == N_PICTURES
    return {int(s.pictures[i]["poc"]): planes for i, planes in e.coded.items()}


def test_single_rank_slots_do_not_change_pictures():
    """Picture slots of one rank change when a picture is coded, never what."""
    base = single_rank_pictures()
    s = schedule.Schedule(N_PICTURES, SUB_GOP, 2, 1, 3)
    e = OraclePictureEngine(s, 0)
    picture_parallel.run_rank(s, 0, e)
    for i, planes in e.coded.items():
        for a, b in zip(planes, base[int(s.pictures[i]["poc"])]):
            assert np.array_equal(a, b)
    # the chain really is hierarchical: a distance-8 reference was used
    p = s.pictures[s.index_of_poc[8]]
    assert p["ref_poc"][0][0] == 0 and p["tid"] == 0


def _worker(rank, world, port, slots, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = schedule.Schedule(N_PICTURES, SUB_GOP, 2, world, slots)
    e = OraclePictureEngine(s, rank, dist)
    picture_parallel.run_rank(s, rank, e)
    dist.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             **{"c%d_%d" % (int(s.pictures[i]["poc"]), c): a
                for i, pl in e.coded.items() for c, a in enumerate(pl)},
             **{"r%d_%d" % (int(s.pictures[i]["poc"]), c): a
                for i, pl in e.received.items() for c, a in enumerate(pl)})
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slots", [(2, 1), (4, 1), (2, 2)])
def test_picture_parallel_gloo(world, slots, tmp_path):
    import torch.multiprocessing as mp
    base = single_rank_pictures()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(world, port, slots, str(tmp_path)), nprocs=world, join=True)
    coded, received = {}, 0
    for rank in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for k in g.files:
            kind, poc, c = k[0], int(k[1:].split("_")[0]), int(k.split("_")[1])
            assert np.array_equal(g[k], base[poc][c]), (rank, k)
            if kind == "c":
                assert (poc, c) not in coded        # every picture coded exactly once
                coded[(poc, c)] = rank
            else:
                received += 1
    assert len(coded) == 3 * N_PICTURES and received > 0
    assert len({r for r in coded.values()}) == world    # every rank coded something


def test_ring_entry_reuse_waits_for_its_writer_and_readers():
    """The picture engine's ring rule (xvc_amd/host/xvc_picture_engine.cc, ClaimEntry):
    before a ring entry is overwritten the new writer waits for the picture that wrote
    its previous content - a picture nobody referenced has no reader events, and its
    slot's stream may still be at work on the entry - and for every reader of it (the
    ordering rule only; no device)."""
    import ctypes as C
    L = schedule.lib()
    idx = np.array([0, 3, 6, 1], np.int32)
    # before the third claim entry 0 has two readers (ids 7 and 9)
    rd = np.zeros((4, 4), np.int32)
    rd[2, :2] = (7, 9)
    ent = np.zeros(4, np.int32)
    waited = np.zeros((4, 8), np.int32)
    L.xvc_host_picture_ring_claims.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4
    assert L.xvc_host_picture_ring_claims(3, 4, idx.ctypes.data, rd.ctypes.data, ent.ctypes.data,
                                          waited.ctypes.data) == 0
    assert ent.tolist() == [0, 0, 0, 1]
    assert waited[0].tolist() == [0] * 8                       # a fresh entry: nothing to wait for
    # picture 3 reuses entry 0, which nobody read: it still waits for picture 0 itself
    assert waited[1].tolist() == [100] + [0] * 7
    # with readers: the writer first, then each of them; the list is cleared
    assert waited[2].tolist() == [100, 7, 9] + [0] * 5
    assert waited[3].tolist() == [0] * 8                       # another entry is untouched
