"""The captured intra SATD pre-selection calls of a real encoder run
(tests/golden/intra_calls_*.npz, tools/gen_intra_golden.py) against the oracle:
for every mode IntraSearch::DetermineSlowIntraModes evaluated, xo_intra_satd_modes
on the same CU, neighbour state and neighbouring reconstruction must return the
encoder's SATD.  (The device's replay: tests/test_gpu_intra_calls.py.)"""
import numpy as np
import pytest

import intra_fixture as ifx
import oracle_intra as oi
import oracle_lib as ol
from xvc_amd import synth


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288), ("c1", 1920, 1080),
                                               ("c0q22", 352, 288), ("c0q37", 352, 288)])
def test_oracle_reproduces_encoder_intra_satd(name, width, height):
    _satd_calls_against_oracle(ifx.load(name), name, width, height, 20000)


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288), ("c1", 1920, 1080)])
def test_oracle_reproduces_the_walked_pictures_intra_satd(name, width, height):
    """tests/golden/intra_order_*.npz (round 5): EVERY DetermineSlowIntraModes call of the
    picture the CU-state walk covers (tests/rd_serial.py kind 4), pinned the same way."""
    _satd_calls_against_oracle(ifx.load_order(name), name, width, height, 9000)


def _satd_calls_against_oracle(fx, name, width, height, least):
    xo = ol.Lib("xo")
    calls, evals, samples = fx["calls"], fx["evals"], fx["samples"]
    assert (np.diff(calls["first_eval"]) == calls["n_eval"][:-1]).all()
    done = 0
    origs = {}
    for ci in range(len(calls)):
        c = calls[ci]
        poc = int(c["poc"])
        if poc not in origs:
            origs[poc] = np.ascontiguousarray(
                synth.SyntheticClip(width, height, 8).frame(poc)[0].astype(np.uint16) << 2)
        x, y, w, h = int(c["x"]), int(c["y"]), int(c["w"]), int(c["h"])
        rec = np.zeros((height + 64, width + 64), np.uint16)
        off, nb = int(c["sample_off"]), int(c["neighbors"])
        if nb & oi.HAS_ABOVE_LEFT:
            rec[y - 1, x - 1] = samples[off]
            off += 1
        if nb & oi.HAS_ABOVE:
            n = w + int(c["above_right"])
            rec[y - 1, x:x + n] = samples[off:off + n]
            off += n
        if nb & oi.HAS_LEFT:
            n = h + int(c["below_left"])
            rec[y:y + n, x - 1] = samples[off:off + n]
        job = np.zeros(1, oi.INTRA_DTYPE)
        job["x"], job["y"], job["w"], job["h"] = x, y, w, h
        job["neighbors"], job["above_right"], job["below_left"] = nb, c["above_right"], c["below_left"]
        orig = np.zeros_like(rec)
        orig[:height, :width] = origs[poc]
        dist = oi.satd_modes(xo, "xo", 10, job, orig, rec)
        e = evals[int(c["first_eval"]):int(c["first_eval"]) + int(c["n_eval"])]
        assert np.array_equal(dist[e["mode"]], e["dist"]), (name, tuple(c))
        done += len(e)
    assert done == len(evals) and done > least


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288),
                                               ("c0q22", 352, 288), ("c0q37", 352, 288)])
def test_oracle_reproduces_encoder_intra_transform_calls(name, width, height):
    """The sampled TransformAndReconstruct calls of intra CUs: the oracle's intra
    prediction from the captured reference samples (CRC equal to the encoder's
    prediction), then xo_residual_pipeline_rdoq with the captured context snapshot:
    count, levels (CRC) and reconstruction (CRC) equal to the encoder's."""
    _itx_calls_against_oracle(ifx.load(name), name, width, height, 2, 1500)


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288)])
def test_oracle_reproduces_the_walked_pictures_intra_transform_calls(name, width, height):
    """tests/golden/intra_order_*.npz: every 5th TransformAndReconstruct call of the walked
    picture's intra CUs (the LM chroma calls' prediction needs the luma rectangle: the GPU
    walk covers them against the capture's counts, levels and distortions)."""
    _itx_calls_against_oracle(ifx.load_order(name), name, width, height, 5, 2000)


def _itx_calls_against_oracle(fx, name, width, height, step, least):
    import ctypes as C
    import oracle_rdoq as orq
    import rd_fixture as rf
    xo = ol.Lib("xo")
    itx, samples = fx["itx"], fx["itx_samples"]
    itx = itx[itx["mode"] <= 66]
    contexts = np.ascontiguousarray(fx["contexts"]).view(orq.RDOQ_CTX_DTYPE).reshape(-1)
    qps = fx["qps"].view(rf.QP_DTYPE).reshape(-1)
    f = xo.dll.xo_residual_pipeline_rdoq
    f.restype = C.c_int
    origs = {}
    done = 0
    for t in itx[::step]:
        poc, comp = int(t["poc"]), int(t["comp"])
        if poc not in origs:
            origs[poc] = [np.ascontiguousarray(p.astype(np.uint16) << 2)
                          for p in synth.SyntheticClip(width, height, 8).frame(poc)]
        ph, pw = origs[poc][comp].shape
        x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
        rec = np.zeros((ph + 64, pw + 64), np.uint16)
        off, nb = int(t["sample_off"]), int(t["neighbors"])
        if nb & oi.HAS_ABOVE_LEFT:
            rec[y - 1, x - 1] = samples[off]
            off += 1
        if nb & oi.HAS_ABOVE:
            n = w + int(t["above_right"])
            rec[y - 1, x:x + n] = samples[off:off + n]
            off += n
        if nb & oi.HAS_LEFT:
            n = h + int(t["below_left"])
            rec[y:y + n, x - 1] = samples[off:off + n]
        job = np.zeros(1, oi.INTRA_DTYPE)
        for k in ("x", "y", "w", "h", "comp", "mode", "neighbors", "above_right", "below_left"):
            job[k] = t[k]
        pred_blk = oi.pred_block(xo, "xo", 10, job, rec, pw, ph)
        assert rf.crc32_rows(pred_blk) == int(t["pred_crc"]), ("prediction", name, tuple(t))
        pred = np.zeros_like(rec)
        pred[y:y + h, x:x + w] = pred_blk
        orig = np.zeros_like(rec)
        orig[:ph, :pw] = origs[poc][comp]
        b = np.zeros(1, ol.TX_DTYPE)
        for k in ("x", "y", "w", "h", "comp", "tx_ver", "qp", "dst4x4"):
            b[k] = t[k]
        b["tx_hor"] = 6 if t["tx_skip"] else t["tx_hor"]
        b["intra_pic"] = 16 | (int(t["scan"]) << 2) | (1 if t["intra_pic"] else 0)
        prm = np.zeros(1, orq.RDOQ_PARAMS_DTYPE)
        q = qps[int(t["qp_index"])]
        prm["lambda"], prm["rd_factor"] = q["lambda"][comp], q["rd_factor"][comp]
        prm["ctx_index"], prm["flags"] = 0, 1          # XVC_RDOQ_INTRA_CU
        ctx = np.ascontiguousarray(contexts[int(t["ctx_index"]):int(t["ctx_index"]) + 1])
        out = np.zeros_like(rec)
        lv = np.zeros((h, w), np.int16)
        vp = lambda a: C.c_void_p(a.ctypes.data)     # noqa: E731
        nnz = f(10, vp(b), vp(ctx), vp(prm), vp(orig), C.c_ssize_t(orig.shape[1]), vp(pred),
                C.c_ssize_t(pred.shape[1]), vp(out), C.c_ssize_t(out.shape[1]), vp(lv))
        assert nnz == int(t["nnz"]), ("count", name, tuple(t), nnz)
        if nnz:
            assert rf.crc32_rows(lv) == int(t["levels_crc"]), ("levels", name, tuple(t))
        if t["completed"]:
            assert rf.crc32_rows(out[y:y + h, x:x + w]) == int(t["rec_crc"]), ("rec", name, tuple(t))
        done += 1
    assert done >= least
