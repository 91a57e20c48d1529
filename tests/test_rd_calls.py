"""The RD-search calls of a real encoder run (tests/golden/rd_calls_tiny.npz,
tools/gen_rd_golden.py) against the CPU oracle: a sample of the bi-prediction
refinement steps and of the affine motion searches the reference encoder made
while coding the small stream fixture - real predictors, bootstrap vectors and
other-list vectors - must come out of the oracle's restatement with the
reference's vectors and distortions.  (The device replays ALL calls of all three
clips: tests/test_gpu_rd_calls.py.)"""
import numpy as np

import oracle_affine_me as oam
import oracle_lib as ol
import rd_fixture as rf
import rd_replay
import stream_fixture as sf

BL = 128


def _pictures():
    fx = sf.StreamFixture("tiny")
    w, h = int(fx.info[0]["width"]), int(fx.info[0]["height"])
    rec = {int(fx.info[i]["poc"]): np.ascontiguousarray(np.pad(fx.planes(i)[0], BL, mode="edge"))
           for i in range(fx.n)}
    return w, h, rec


def test_fixture_tables_are_consistent():
    for name in ("tiny", "c0", "c1", "c0q22", "c0q37"):
        rd = rf.load(name)
        calls, evals = rd["calls"], rd["evals"]
        assert len(calls) > 50000 and len(evals) > 10000
        assert calls["eval"].min() >= 0 and calls["eval"].max() < len(evals)
        assert np.all(np.diff(calls["eval"]) >= 0)          # capture order
        assert evals["ctx_index"].max() < len(rd["contexts"])
        assert evals["qp_index"].max() < len(rd["qps"])
        lic = (evals["flags"] & rf.FLAG_LIC) != 0
        assert np.all(evals["nb_index"][lic] >= 0) and np.all(evals["nb_index"][~lic] < 0)
        nb = rd["neighbours"]
        if len(nb):
            assert int((nb["sample_off"] + nb["sample_count"]).max()) == len(rd["nb_samples"])
        st = rd["steps"]
        assert set(np.unique(st["kind"]).tolist()) <= {1, 2, 3}
        m = rd["merges"]
        assert np.all(np.sort(m["order"], axis=1) == np.arange(5)[None, :])
        assert np.all(np.diff(m["cost"], axis=1) >= 0)


def test_oracle_reproduces_encoder_bipred_steps():
    w, h, rec = _pictures()
    st = rf.load("tiny")["steps"]
    st = st[(st["kind"] == rf.KIND_BI) & ((st["flags"] & rf.FLAG_LIC) == 0)]
    assert len(st) > 10000
    xo = ol.Lib("xo")
    orig = {}
    sizes = set()
    for s in st[::4]:
        poc = int(s["poc"])
        if poc not in orig:
            orig[poc] = rd_replay.original_planes(w, h, poc)[0]
        job = ol.BiBlock()
        b = job.blk
        b.x, b.y, b.w, b.h = int(s["x"]), int(s["y"]), int(s["w"]), int(s["h"])
        b.fullpel_mv = 1 if s["flags"] & rf.FLAG_FULLPEL else 0
        k = int(s["start_mvp_idx"])
        b.mvp_x, b.mvp_y = int(s["mvp"][k, 0, 0]), int(s["mvp"][k, 0, 1])
        b.lambda16, b.search_range = int(s["lambda16"]), 4
        job.other_mv_x, job.other_mv_y = int(s["other_mv"][0, 0]), int(s["other_mv"][0, 1])
        job.boot_mv_x, job.boot_mv_y = int(s["boot"][0, 0]), int(s["boot"][0, 1])
        mv, dist = xo.bipred_search(10, job, w, h, orig[poc], rec[int(s["other_ref_poc"])],
                                    rec[int(s["ref_poc"])], BL)
        assert mv == (int(s["mv"][0, 0]), int(s["mv"][0, 1])) and dist == int(s["dist"]), tuple(s)
        sizes.add((int(s["w"]), int(s["h"])))
    assert len(sizes) >= 10


def test_oracle_reproduces_encoder_affine_searches():
    w, h, rec = _pictures()
    st = rf.load("tiny")["steps"]
    st = st[st["kind"] != rf.KIND_BI]
    assert len(st) > 1000
    xo = ol.Lib("xo")
    orig = {}
    kinds = set()
    for s in st:
        poc = int(s["poc"])
        if poc not in orig:
            orig[poc] = rd_replay.original_planes(w, h, poc)[0]
        blk = np.zeros(1, oam.BLOCK_DTYPE)[0]
        blk["x"], blk["y"], blk["w"], blk["h"] = s["x"], s["y"], s["w"], s["h"]
        bipred = s["kind"] == rf.KIND_AFFINE_BI
        blk["flags"] = ((oam.HAS_BOOTSTRAP if s["flags"] & rf.FLAG_HAS_BOOT else 0) |
                        (oam.BIPRED if bipred else 0))
        blk["lambda16"] = s["lambda16"]
        blk["mvp"] = s["mvp"][int(s["start_mvp_idx"])]
        blk["bootstrap"] = s["boot"]
        blk["other_mv"] = s["other_mv"]
        other = rec[int(s["other_ref_poc"])] if bipred else None
        r = oam.affine_me(xo, 10, blk, w, h, orig[poc], rec[int(s["ref_poc"])], BL, other)
        assert np.array_equal(r["mv"], s["mv"]) and int(r["dist"]) == int(s["dist"]), tuple(s)
        kinds.add(int(s["kind"]))
    assert kinds == {2, 3}
