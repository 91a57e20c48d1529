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


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288), ("c1", 1920, 1080),
                                               ("c0q22", 352, 288), ("c0q37", 352, 288)])
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


@pytest.mark.parametrize("name,width,height", [("tiny", 136, 72), ("c0", 352, 288),
                                               ("c1", 1920, 1080), ("c0q22", 352, 288),
                                               ("c0q37", 352, 288)])
def test_intra_transform_and_reconstruct_of_a_real_encode(gpu, name, width, height):
    """A sample of the TransformAndReconstruct calls the encoder made for INTRA CUs
    (transform_encoder.cc:203-285): prediction from the captured reference samples
    (xvcgpu_intra_pred_batch; its CRC is checked), then xvcgpu_residual_rdoq_batch
    with the CABAC context snapshot RdoQuant::QuantRdo read at that moment - all
    three scan orders, transform skip, the transform-select pairs, luma and chroma
    (two trees in intra pictures) - against the encoder's count, levels (CRC),
    reconstruction (CRC) and returned distortion."""
    import rd_fixture as rf
    api, ctx = gpu
    fx = ifx.load(name)
    if "itx" not in fx:
        pytest.skip("no intra transform calls in the fixture")
    itx, samples = fx["itx"], fx["itx_samples"]
    contexts = fx["contexts"].view(api.RDOQ_CTX_DTYPE).reshape(-1)
    qps = fx["qps"].view(rf.QP_DTYPE).reshape(-1)
    assert len(itx) >= 3000 and (itx["scan"] == 1).sum() > 100 and (itx["scan"] == 2).sum() > 100
    rec_nb, pred, rec_out = (ctx.picture(width, height, 10) for _ in range(3))
    planes = [np.zeros((height + 2 * BL, width + 2 * BL), np.uint16),
              np.zeros((height // 2 + BL, width // 2 + BL), np.uint16),
              np.zeros((height // 2 + BL, width // 2 + BL), np.uint16)]
    views = [p[(BL >> (1 if c else 0)):, (BL >> (1 if c else 0)):] for c, p in enumerate(planes)]
    done = bad = bad_pred = 0
    for poc in np.unique(itx["poc"]):
        O = ctx.picture(width, height, 10)
        O.upload(rd_replay.original_planes(width, height, int(poc)), BL)
        for comp in range(3):
            idx = np.flatnonzero((itx["poc"] == poc) & (itx["comp"] == comp))
            if not len(idx):
                continue
            # footprint in luma units (the layering works on a 4x4 luma grid)
            s = 1 if comp else 0
            foot = np.zeros(len(idx), ifx.CALL_DTYPE)
            foot["x"], foot["y"] = itx["x"][idx].astype(np.int32) << s, itx["y"][idx].astype(np.int32) << s
            foot["w"] = np.minimum(itx["w"][idx].astype(np.int32) << s, 255)
            foot["h"] = np.minimum(itx["h"][idx].astype(np.int32) << s, 255)
            foot["above_right"] = np.minimum(itx["above_right"][idx].astype(np.int32) << s, 255)
            foot["below_left"] = np.minimum(itx["below_left"][idx].astype(np.int32) << s, 255)
            lay = _layers(foot)
            for k in range(int(lay.max()) + 1):
                sel = idx[lay == k]
                c = itx[sel]
                jobs = np.zeros(len(sel), api.INTRA_DTYPE)
                for j, t in enumerate(c):
                    x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
                    off, nb = int(t["sample_off"]), int(t["neighbors"])
                    v = views[comp]
                    if nb & api.INTRA_HAS_ABOVE_LEFT:
                        v[y - 1, x - 1] = samples[off]
                        off += 1
                    if nb & api.INTRA_HAS_ABOVE:
                        n = w + int(t["above_right"])
                        v[y - 1, x:x + n] = samples[off:off + n]
                        off += n
                    if nb & api.INTRA_HAS_LEFT:
                        n = h + int(t["below_left"])
                        v[y:y + n, x - 1] = samples[off:off + n]
                for f in ("x", "y", "w", "h", "comp", "mode", "neighbors", "above_right", "below_left"):
                    jobs[f] = c[f]
                rec_nb.upload(planes, BL)
                ctx.intra_pred_batch(rec_nb, pred, jobs)
                blocks = np.zeros(len(sel), api.TX_DTYPE)
                for f in ("x", "y", "w", "h", "comp", "tx_ver", "qp"):
                    blocks[f] = c[f]
                blocks["tx_hor"] = np.where(c["tx_skip"] != 0, 6, c["tx_hor"])
                blocks["dst4x4"] = c["dst4x4"]
                blocks["intra_pic"] = (api.TXF_RDOQ | (c["scan"].astype(np.int64) << api.TXF_SCAN_SHIFT) |
                                       np.where(c["intra_pic"] != 0, 1, 0))
                uctx, inv = np.unique(c["ctx_index"], return_inverse=True)
                prm = np.zeros(len(sel), api.RDOQ_PARAMS_DTYPE)
                q = qps[c["qp_index"]]
                prm["lambda"] = q["lambda"][np.arange(len(sel)), comp]
                prm["rd_factor"] = q["rd_factor"][np.arange(len(sel)), comp]
                prm["ctx_index"] = inv
                prm["flags"] = api.RDOQ_INTRA_CU
                levels, off_, nnz = ctx.residual_rdoq_batch(O, pred, rec_out, blocks, contexts[uctx], prm)
                cands = np.zeros(len(sel), api.CAND_DTYPE)
                for f in ("x", "y", "w", "h"):
                    cands[f] = blocks[f]
                cands["metric"] = 7 if comp == 0 else 0
                cands["qp"] = c["qp_luma"]
                dist = np.zeros(len(sel), np.uint64)
                for qi in np.unique(c["qp_index"]):
                    mm = np.flatnonzero(c["qp_index"] == qi)
                    dist[mm] = ctx.metric_batch(O, rec_out, comp, cands[mm],
                                                weight=float(qps["dist_weight"][qi, comp]))
                pp, rp = pred.download()[comp], rec_out.download()[comp]
                for j, t in enumerate(c):
                    x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
                    if rf.crc32_rows(pp[y:y + h, x:x + w]) != int(t["pred_crc"]):
                        bad_pred += 1
                        continue
                    lv = levels[int(off_[j]):int(off_[j]) + w * h]
                    good = int(nnz[j]) == int(t["nnz"]) and (
                        t["nnz"] == 0 or rf.crc32_rows(lv) == int(t["levels_crc"]))
                    if good and t["completed"]:
                        good = (rf.crc32_rows(rp[y:y + h, x:x + w]) == int(t["rec_crc"]) and
                                int(dist[j]) == int(t["dist"]))
                    if not good and bad < 5:
                        print("mismatch", name, tuple(t), int(nnz[j]), int(dist[j]))
                    bad += 0 if good else 1
                    done += 1
        O.destroy()
    for p in (rec_nb, pred, rec_out):
        p.destroy()
    assert bad_pred == 0 and bad == 0 and done == len(itx), (done, bad, bad_pred)
