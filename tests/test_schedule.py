"""The picture-level schedule (xvc_amd/host/xvc_picture_schedule.cc): sub-GOP
arithmetic against the reference's tables (golden + live when oracle/_ref is
built), reference lists against what the reference encoder put into the stream
fixtures, and the properties of the played ThreadEncoder schedule."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import stream_fixture as sf
from xvc_amd import schedule

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_subgop_arithmetic_golden():
    rows = np.load(os.path.join(GOLDEN, "subgop.npz"))["rows"]
    assert len(rows) == 7 * 131
    for length, n, doc, poc, tid in rows:
        assert schedule.doc_from_poc(int(n), int(length)) == doc, (length, n)
        assert schedule.poc_from_doc(int(n), int(length)) == poc, (length, n)
        assert schedule.tid_from_doc(int(n), int(length)) == tid, (length, n)
    assert schedule.doc_from_poc(3, 12) == -1       # a length that is not restated


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
def test_subgop_arithmetic_vs_reference():
    xr = ol.Lib("xr")
    for length in (1, 2, 4, 8, 16, 32, 64):
        for n in range(0, 400):
            assert schedule.doc_from_poc(n, length) == xr.dll.xr_doc_from_poc(n, length)
            assert schedule.poc_from_doc(n, length) == xr.dll.xr_poc_from_doc(n, length)
            assert schedule.tid_from_doc(n, length) == xr.dll.xr_tid_from_doc(n, length)


@pytest.mark.parametrize("name,length", [("tiny", 4), ("c0", 16), ("c1", 4)])
def test_reference_lists_match_the_reference_encoder(name, length):
    """poc / doc / tid and both reference lists of every picture of the streams
    the reference encoder produced (ReferenceListSorter::Prepare, 2 per list)."""
    g = np.load(os.path.join(GOLDEN, "stream_%s.npz" % name))
    info = g["info"].view(sf.STREAM_INFO_DTYPE)
    s = schedule.Schedule(len(info), sub_gop_length=length, num_ref_pics=2)
    assert len(s.pictures) == len(info)
    for p, r in zip(s.pictures, info):      # both in coding order
        assert (p["poc"], p["doc"], p["tid"]) == (r["poc"], r["doc"], r["tid"])
        assert p["num_ref"].tolist() == r["num_ref"].tolist(), int(p["poc"])
        assert p["ref_poc"].tolist() == r["ref_poc"].tolist(), int(p["poc"])
        assert bool(p["intra"]) == (r["pic_type"] == 2)
        # highest-layer pictures are never references (picture_encoder.cc:149)
        if r["highest_layer"] and r["pic_type"] != 2:
            assert not p["is_reference"]


@pytest.mark.parametrize("ranks,slots", [(1, 1), (1, 3), (2, 1), (2, 3), (4, 2), (8, 3)])
def test_schedule_properties(ranks, slots):
    n = 1 + 16 * 6
    s = schedule.Schedule(n, 16, 2, ranks, slots)
    P = s.pictures
    by_poc = s.index_of_poc
    assert sorted(P["poc"].tolist()) == list(range(n))
    # every dependency finished before its consumer starts; one picture per worker at a time
    for p in P:
        for l in range(2):
            for k in range(p["num_ref"][l]):
                d = P[by_poc[int(p["ref_poc"][l][k])]]
                assert d["finish"] <= p["start"] and d["doc"] < p["doc"]
        assert p["rank"] == p["worker"] % ranks and p["slot"] == p["worker"] // ranks
    for w in range(ranks * slots):
        mine = sorted(P[P["worker"] == w], key=lambda q: q["start"])
        for a, b in zip(mine, mine[1:]):
            assert a["finish"] <= b["start"]
    # the policy: when a picture started, no ready picture of a lower layer was waiting
    # (among the pictures queued then: `window` places from the oldest unfinished one)
    for p in P:
        first_open = min(i for i, q in enumerate(P) if q["finish"] > p["start"])
        for qi, q in enumerate(P):
            if q["start"] > p["start"] and q["tid"] < p["tid"] and qi < first_open + s.window:
                deps = [P[by_poc[int(x)]] for l in range(2) for x in q["ref_poc"][l][:q["num_ref"][l]]]
                assert any(d["finish"] > p["start"] for d in deps), (int(p["poc"]), int(q["poc"]))
    # the timeline: transfers exactly for (reference, consumer rank) pairs across ranks,
    # after the encode, before the first consumer on that rank
    ops = s.ops
    enc_pos = {int(o["picture"]): i for i, o in enumerate(ops) if o["kind"] == schedule.ENCODE}
    assert len(enc_pos) == n
    need = set()
    for i, p in enumerate(P):
        for l in range(2):
            for k in range(p["num_ref"][l]):
                j = by_poc[int(p["ref_poc"][l][k])]
                if P[j]["rank"] != p["rank"]:
                    need.add((j, int(p["rank"])))
    got = {(int(o["picture"]), int(o["dst_rank"])) for o in ops if o["kind"] == schedule.TRANSFER}
    assert got == need
    if ranks == 1:
        assert not got
    for pos, o in enumerate(ops):
        if o["kind"] != schedule.TRANSFER:
            continue
        assert o["src_rank"] == P[o["picture"]]["rank"] and pos > enc_pos[int(o["picture"])]
        for i, p in enumerate(P):
            if p["rank"] == o["dst_rank"] and int(P[o["picture"]]["poc"]) in \
                    [int(x) for l in range(2) for x in p["ref_poc"][l][:p["num_ref"][l]]]:
                assert enc_pos[i] > pos
    # more workers never take longer; the single worker takes one unit per picture
    if ranks * slots == 1:
        assert s.makespan == n
    assert s.makespan <= schedule.Schedule(n, 16, 2, 1, 1).makespan
    assert s.window == 2 + 16 * ranks * slots + 1


def test_schedule_speedup_matches_the_layer_structure():
    """Sub-GOP 16: layers of 1, 1, 2, 4, 8 pictures.  With unit picture times 8
    workers cannot beat the 5-picture critical path of one sub-GOP by more than
    the overlap of successive sub-GOPs allows (16 / 5 = 3.2x per
    sub-GOP, more when sub-GOPs are pipelined)."""
    n = 1 + 16 * 8
    t1 = schedule.Schedule(n, 16, 2, 1, 1).makespan
    t8 = schedule.Schedule(n, 16, 2, 8, 1).makespan
    t2 = schedule.Schedule(n, 16, 2, 2, 1).makespan
    assert t1 == n and t2 <= (n + 1) // 2 + 4 * 8
    assert 3.2 <= t1 / t8 <= 8.0


def test_run_walks_only_own_entries():
    s = schedule.Schedule(33, 16, 2, ranks=2, slots_per_rank=1)
    for rank in range(2):
        log = []
        s.run(rank, lambda p, i: log.append(("E", i)), lambda p, i, r: log.append(("S", i, r)),
              lambda p, i, r: log.append(("R", i, r)))
        exp = []
        for o in s.ops:
            if o["kind"] == schedule.ENCODE and o["src_rank"] == rank:
                exp.append(("E", int(o["picture"])))
            elif o["kind"] == schedule.TRANSFER and o["src_rank"] == rank:
                exp.append(("S", int(o["picture"]), int(o["dst_rank"])))
            elif o["kind"] == schedule.TRANSFER and o["dst_rank"] == rank:
                exp.append(("R", int(o["picture"]), int(o["src_rank"])))
        assert log == exp and len(log) > 10
    with pytest.raises(ZeroDivisionError):
        s.run(0, lambda p, i: 1 // 0)
