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


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288), ("c1", 1920, 1080)])
def test_oracle_reproduces_encoder_intra_satd(name, width, height):
    xo = ol.Lib("xo")
    fx = ifx.load(name)
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
    assert done == len(evals) and done > 20000
