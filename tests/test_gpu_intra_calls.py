"""The intra search's SATD pre-selection on the calls a real encoder run made
(tests/golden/intra_calls_*.npz, tools/gen_intra_golden.py): a sample of the
reference encoder's IntraSearch::DetermineSlowIntraModes calls (intra_search.cc:
188-305) - CUs 4x4 ... 64x64 of every picture of the clip, with the neighbour
state DetermineNeighbors reported and the reconstruction's row above / column to
the left at that moment of the RD search - replayed through
xvcgpu_intra_satd_batch: for every mode the encoder evaluated, the device's SATD
must be the encoder's."""
import numpy as np
import pytest

import intra_fixture as ifx
import rd_replay

pytestmark = pytest.mark.gpu
BL = rd_replay.BL


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def _layers(calls):
    """Calls whose reference samples (row above incl. above-right, column left
    incl. below-left, the corner) do not share a 4x4 cell go into one layer."""
    x0 = np.maximum(calls["x"].astype(np.int64) // 4 - 1, 0)
    y0 = np.maximum(calls["y"].astype(np.int64) // 4 - 1, 0)
    x1 = (calls["x"].astype(np.int64) + calls["w"] + calls["above_right"] + 3) // 4
    y1 = (calls["y"].astype(np.int64) + calls["h"] + calls["below_left"] + 3) // 4
    last = np.full((int(y1.max()) + 1, int(x1.max()) + 1), -1, np.int32)
    out = np.zeros(len(calls), np.int32)
    for i in range(len(calls)):
        v = last[y0[i]:y1[i], x0[i]:x1[i]]
        k = int(v.max()) + 1
        v[...] = k
        out[i] = k
    return out


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288), ("c1", 1920, 1080)])
def test_intra_satd_preselection_of_a_real_encode(gpu, name, width, height):
    api, ctx = gpu
    fx = ifx.load(name)
    calls, evals, samples = fx["calls"], fx["evals"], fx["samples"]
    assert len(calls) > 300 and len(evals) > 10000
    rec = ctx.picture(width, height, 10)
    plane = np.zeros((height + 2 * BL, width + 2 * BL), np.uint16)
    chroma = [np.zeros((height // 2 + BL, width // 2 + BL), np.uint16)] * 2
    view = plane[BL:, BL:]
    done = bad = n_layers = 0
    for poc in np.unique(calls["poc"]):
        O = ctx.picture(width, height, 10)
        O.upload(rd_replay.original_planes(width, height, int(poc)), BL)
        idx = np.flatnonzero(calls["poc"] == poc)
        lay = _layers(calls[idx])
        for k in range(int(lay.max()) + 1):
            sel = idx[lay == k]
            jobs = np.zeros(len(sel), api.INTRA_DTYPE)
            for j, ci in enumerate(sel):
                c = calls[ci]
                x, y, w, h = int(c["x"]), int(c["y"]), int(c["w"]), int(c["h"])
                off = int(c["sample_off"])
                nb = int(c["neighbors"])
                if nb & api.INTRA_HAS_ABOVE_LEFT:
                    view[y - 1, x - 1] = samples[off]
                    off += 1
                if nb & api.INTRA_HAS_ABOVE:
                    n = w + int(c["above_right"])
                    view[y - 1, x:x + n] = samples[off:off + n]
                    off += n
                if nb & api.INTRA_HAS_LEFT:
                    n = h + int(c["below_left"])
                    view[y:y + n, x - 1] = samples[off:off + n]
                    off += n
                jb = jobs[j]
                jb["x"], jb["y"], jb["w"], jb["h"], jb["comp"] = x, y, w, h, 0
                jb["neighbors"] = nb
                jb["above_right"], jb["below_left"] = c["above_right"], c["below_left"]
                jobs[j] = jb
            rec.upload([plane] + chroma, BL)
            dist = ctx.intra_satd_batch(O, rec, jobs)
            for j, ci in enumerate(sel):
                c = calls[ci]
                e = evals[int(c["first_eval"]):int(c["first_eval"]) + int(c["n_eval"])]
                assert (e["call"] == ci).all()
                got = dist[j, e["mode"]]
                miss = got != e["dist"]
                bad += int(miss.sum())
                done += len(e)
                if miss.any() and bad <= 5:
                    m = int(np.flatnonzero(miss)[0])
                    print("mismatch", name, tuple(c), "mode", int(e["mode"][m]), int(got[m]),
                          int(e["dist"][m]))
            n_layers += 1
        O.destroy()
    rec.destroy()
    assert done == len(evals) and bad == 0, (done, bad, n_layers)
