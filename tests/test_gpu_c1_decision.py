"""C1, decision half on the device (xvcgpu_tx_eval_batch / xvcgpu_root_cbf_batch,
csrc/k_rd.h) against the reference's own InterSearch::CompressAndEvalCbf
(inter_search.cc:261-365) and TransformEncoder::CompressAndEvalTransform
(transform_encoder.cc:53-201), for CUs 8x8 ... 64x64.

For every CU the device evaluates every alternative the reference tries - default
transform, transform skip (blocks of at most 16 samples), the four transform-select
pairs (luma), all with RdoQuant on the picture-initial contexts - and their
distortions; the bits of each alternative are asked of the reference's entropy
coder (xr_c1_bits: the host input of this stage); the device folds them (first
pass, root-cbf-zero test, second-pass gate, second pass).  cbf flags, root cbf,
transform-select index, transform-skip flags, the returned distortion and the
reconstruction must equal what CompressAndEvalCbf decided on the same CU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import rd_fixture as rf

pytestmark = pytest.mark.gpu
BL = 128


@pytest.fixture(scope="module")
def gpu():
    from xvc_amd import api
    ctx = api.Context(0)
    yield api, ctx
    ctx.close()


def _pictures(rng, bd, pw, ph):
    """orig / ref0 / ref1: three padded planes each; the original is the mean of
    ref0 displaced by (-4, 2) and ref1 displaced by (4, -2) samples, plus noise."""
    mx = (1 << bd) - 1

    def tex(h, w, amp):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        t = (np.sin(xx / 7.0) * np.cos(yy / 9.0) * 0.3 + np.sin((xx + 2 * yy) / 23.0) * 0.2 + 0.5)
        return np.clip(t * mx + rng.integers(-amp, amp + 1, (h, w)), 0, mx)

    planes = {}
    for name, amp in (("ref0", 2), ("ref1", 3)):
        planes[name] = [tex(ph + 2 * BL, pw + 2 * BL, amp).astype(np.uint16),
                        tex(ph // 2 + BL, pw // 2 + BL, amp // 2).astype(np.uint16),
                        tex(ph // 2 + BL, pw // 2 + BL, amp // 2).astype(np.uint16)]
    orig = []
    for c in range(3):
        s = 1 if c else 0
        a = np.roll(planes["ref0"][c].astype(np.int32), (-2 >> s, 4 >> s), (0, 1))
        b = np.roll(planes["ref1"][c].astype(np.int32), (2 >> s, -4 >> s), (0, 1))
        o = (a + b + 1) // 2 + rng.integers(-1, 2, a.shape)
        # the left half of the picture: strong local changes (blocks that want
        # coefficients, some of them sharp: transform skip / other transform types);
        # the right half: what the prediction already gives (all-zero blocks win)
        for _ in range(160 if c == 0 else 500):
            y0 = rng.integers(0, a.shape[0] - 8)
            x0 = rng.integers(0, a.shape[1] // 2 - 8)
            sz = int(rng.choice([1, 1, 2, 3, 6]))
            o[y0:y0 + sz, x0:x0 + sz] += rng.integers(-120, 120) * (3 if sz == 1 else 1)
        orig.append(np.clip(o, 0, mx).astype(np.uint16))
    return orig, planes["ref0"], planes["ref1"]


def _ptrs(planes):
    p = (C.c_void_p * 3)(*[pl[(BL >> (1 if c else 0)):, (BL >> (1 if c else 0)):].ctypes.data
                           for c, pl in enumerate(planes)])
    s = np.array([pl.strides[0] // 2 for pl in planes], np.int64)
    return p, s


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
def test_compress_and_eval_cbf_vs_reference(gpu):
    total = {}
    for bd, qp, fast_select in [(10, 32, 1), (10, 27, 1), (8, 22, 1), (10, 37, 0), (10, 27, 0)]:
        st = _run(gpu, bd, qp, fast_select)
        for k, v in st.items():
            total[k] = total.get(k, 0) + v
    # a second content (the seed of soak round 47) on which transform skip - evaluated
    # for the 4x4 chroma blocks of the 8-wide CUs - wins as well
    tskip = 0
    for bd, qp, fast_select in [(10, 32, 1), (10, 27, 1), (8, 22, 1), (10, 37, 0), (10, 27, 0)]:
        tskip += _run(gpu, bd, qp, fast_select, seed_shift=7919 * 47)["tskip"]
    # input coverage (the parity assertions are in _run): at the committed seed the content
    # makes a transform-select pair win at least twice; a soak run's shifted seed need not
    import os
    soak = int(os.environ.get("XVC_SOAK", "0")) != 0
    assert total["cbf"] >= 20 and total["root0"] >= 10 and (soak or total["sel"] >= 2), total
    assert soak or tskip >= 1, tskip


def _run(gpu, bd, qp, fast_select, seed_shift=0):
    api, ctx = gpu
    xr = C.CDLL(ol.REF_SO)
    xr.xr_c1_create.restype = C.c_void_p
    xr.xr_c1_create.argtypes = [C.c_int] * 4 + [C.c_double] + [C.c_int] * 6 + [C.c_void_p] * 7
    xr.xr_c1_destroy.argtypes = [C.c_void_p]
    xr.xr_c1_qp.argtypes = [C.c_void_p] * 3
    xr.xr_c1_reference.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    xr.xr_c1_bits.restype = C.c_uint32
    xr.xr_c1_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7700 + bd + qp + seed_shift)
    pw, ph = 512, 256
    orig, ref0, ref1 = _pictures(rng, bd, pw, ph)
    O, R0, R1, P, Rc = (ctx.picture(pw, ph, bd) for _ in range(5))
    O.upload(orig, BL)
    R0.upload(ref0, BL)
    R1.upload(ref1, BL)
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    # non-overlapping CUs on a 64-grid, all shapes 8..64
    cus = []
    for gy in range(0, ph, 64):
        for gx in range(0, pw, 64):
            cus.append((gx, gy, int(rng.choice([8, 8, 16, 32, 64])),
                        int(rng.choice([8, 8, 16, 32, 64]))))
    n = len(cus)
    inter_dir = rng.integers(0, 3, n)
    merge = rng.integers(0, 2, n)
    mv = np.zeros((n, 4), np.int32)
    # the true motion, every third CU off by less than half a sample
    off = np.where((np.arange(n) % 3 == 0)[:, None], rng.integers(-6, 7, (n, 2)), 0)
    mv[:, 0:2] = np.array([-64, 32]) + off
    mv[:, 2:4] = np.array([64, -32]) - off
    po, so = _ptrs(orig)
    p0, s0 = _ptrs(ref0)
    p1, s1 = _ptrs(ref1)
    envs, qpi, wts = [], np.zeros((n, 9), np.int64), np.zeros((n, 3), np.float64)
    for i, (x, y, w, h) in enumerate(cus):
        m = np.ascontiguousarray(mv[i])
        e = xr.xr_c1_create(bd, pw, ph, qp, lam, x, y, w, h, int(inter_dir[i]), int(merge[i]),
                            m.ctypes.data, po, so.ctypes.data, p0, s0.ctypes.data, p1,
                            s1.ctypes.data)
        envs.append(e)
        xr.xr_c1_qp(e, qpi[i].ctypes.data, wts[i].ctypes.data)
    # ---- the reference's decisions (best_cu_cost: never / always / sometimes gating)
    best_cu_cost = np.array([[0xffffffffffffffff, 0, 9000][i % 3] for i in range(n)], np.uint64)
    exp = np.zeros((n, 13), np.int64)
    for i in range(n):
        xr.xr_c1_reference(envs[i], int(best_cu_cost[i]), fast_select, exp[i].ctypes.data)

    # ---- device: prediction of every CU, cbf-zero distortions
    jobs = np.zeros((n, 3), api.INTER_DTYPE)
    for c in range(3):
        j = jobs[:, c]
        j["x"], j["y"] = [u[0] for u in cus], [u[1] for u in cus]
        j["w"], j["h"], j["comp"] = [u[2] for u in cus], [u[3] for u in cus], c
        j["ref"][:, 0] = np.where(inter_dir != 1, 0, -1)
        j["ref"][:, 1] = np.where(inter_dir != 0, 1, -1)
        j["mv"][:, 0, 0, :] = mv[:, 0:2]
        j["mv"][:, 1, 0, :] = mv[:, 2:4]
    ctx.inter_pred_batch([R0, R1], P, P, jobs.reshape(-1))
    from xvc_amd import pipeline
    init_ctx = pipeline.rdoq_init_contexts(qp, 0)      # PicturePredictionType::kBi

    def metric(pic, comp, idx):
        cands = np.zeros(len(idx), api.CAND_DTYPE)
        s = 1 if comp else 0
        for f, k in (("x", 0), ("y", 1), ("w", 2), ("h", 3)):
            cands[f] = [cus[i][k] >> s for i in idx]
        cands["metric"] = 7 if comp == 0 else 0
        cands["qp"] = qp
        out = np.zeros(len(idx), np.uint64)
        for wv in np.unique(wts[idx, comp]):
            m = np.flatnonzero(wts[idx, comp] == wv)
            out[m] = ctx.metric_batch(O, pic, comp, cands[m], weight=float(wv))
        return out

    all_idx = np.arange(n)
    dist_zero = np.stack([metric(P, c, all_idx) for c in range(3)], 1)

    def evaluate(comp, idx, tx_hor, tx_ver, skip):
        """TransformAndReconstruct of component `comp` of CUs idx with one transform
        choice -> (levels list, nnz, dist_reco, planes of the reconstruction)."""
        s = 1 if comp else 0
        b = np.zeros(len(idx), api.TX_DTYPE)
        for f, k in (("x", 0), ("y", 1), ("w", 2), ("h", 3)):
            b[f] = [cus[i][k] >> s for i in idx]
        b["comp"] = comp
        b["tx_hor"] = 6 if skip else tx_hor
        b["tx_ver"] = tx_ver
        b["qp"] = qpi[idx, comp]
        b["intra_pic"] = api.TXF_RDOQ
        prm = np.zeros(len(idx), api.RDOQ_PARAMS_DTYPE)
        prm["lambda"], prm["rd_factor"] = qpi[idx, 3 + comp], qpi[idx, 6 + comp]
        levels, off, nnz = ctx.residual_rdoq_batch(O, P, Rc, b, init_ctx, prm)
        lv = [levels[int(off[k]):int(off[k]) + int(b["w"][k]) * int(b["h"][k])].copy()
              for k in range(len(idx))]
        return lv, nnz, metric(Rc, comp, idx), Rc.download()

    def bits(i, kind, comp, cbf, tskip, sel, root, levels):
        st = np.array(list(cbf) + list(tskip) + [sel, root], np.int32)
        keep = [np.ascontiguousarray(l, np.int16) if l is not None else None for l in levels]
        lp = (C.c_void_p * 3)(*[k.ctypes.data if k is not None else None for k in keep])
        return int(xr.xr_c1_bits(envs[i], kind, comp, st.ctypes.data, lp))

    # TransformType pairs of the select indices (CodingUnit::SetTransformFromSelectIdx,
    # coding_unit.cc: kDct8 / kDst7 by the two bits of the index - asked of the stream
    # fixtures' convention: vertical = bit 1, horizontal = bit 0)
    sel_types = rf_select_types()
    state = [dict(cbf=[0, 0, 0], tskip=[0, 0, 0], sel=-1, levels=[None] * 3, reco=[0] * 3,
                  resi=[0] * 3, cost=[0] * 3, rec=[None] * 3) for _ in range(n)]
    pred_planes = P.download()
    alt_store = {}

    def run_pass(comp, idx, kinds, first_pass):
        """One CompressAndEvalTransform per CU of idx for component comp."""
        alts, per = [], {i: [] for i in idx}
        for kind, sel in kinds:
            if kind == api.TXE_KIND_TSKIP:
                sub = [i for i in idx if (cus[i][2] >> (1 if comp else 0)) *
                       (cus[i][3] >> (1 if comp else 0)) <= 16]
            else:
                sub = list(idx)
            if not sub:
                continue
            # (no select index: kDct2 both ways, coding_unit.cc:399-404)
            th, tv = (sel_types[sel] if kind == api.TXE_KIND_SELECT else (1, 1))
            lv, nnz, dist, planes = evaluate(comp, np.array(sub), th, tv, kind == api.TXE_KIND_TSKIP)
            for k, i in enumerate(sub):
                cbf = int(nnz[k] != 0)
                # the signalling invariants TransformAndReconstruct enforces (:239-252)
                invalid = (kind == api.TXE_KIND_SELECT and comp == 0 and not cbf) or \
                          (kind == api.TXE_KIND_TSKIP and not cbf)
                st = state[i]
                c_cbf, c_ts = list(st["cbf"]), list(st["tskip"])
                c_cbf[comp], c_ts[comp] = cbf, int(kind == api.TXE_KIND_TSKIP)
                lvls = list(st["levels"])
                lvls[comp] = lv[k]
                b = 0 if invalid else bits(i, 0, comp, c_cbf, c_ts,
                                           sel if kind == api.TXE_KIND_SELECT else -1, 1, lvls)
                s = 1 if comp else 0
                x, y, w, h = (v >> s for v in cus[i])
                per[i].append(dict(kind=kind, sel=sel, cbf=cbf, dist=int(dist[k]), bits=b,
                                   invalid=invalid, levels=lv[k],
                                   rec=planes[comp][y:y + h, x:x + w].copy()))
        jobs_a = np.zeros(len(idx), api.TXE_JOB_DTYPE)
        flat = []
        for q, i in enumerate(idx):
            jobs_a[q]["lambda"] = lam
            jobs_a[q]["prev_cost"] = 0xffffffffffffffff if first_pass else state[i]["cost"][comp]
            jobs_a[q]["dist_zero"] = dist_zero[i, comp]
            jobs_a[q]["bits_zero"] = bits(i, 1, comp, state[i]["cbf"], state[i]["tskip"], -1, 1,
                                          [None] * 3)
            jobs_a[q]["alt_first"], jobs_a[q]["n_alt"] = len(flat), len(per[i])
            jobs_a[q]["flags"] = ((api.TXE_CBF_ZERO | (api.TXE_FAST_SELECT if fast_select else 0))
                                  if first_pass else
                                  (api.TXE_PREV_CBF if state[i]["cbf"][comp] else 0))
            flat += per[i]
        arr = np.zeros(len(flat), api.TXE_ALT_DTYPE)
        for k, a in enumerate(flat):
            arr[k]["dist_reco"] = api.TXE_DIST_INVALID if a["invalid"] else a["dist"]
            arr[k]["dist_resi"] = arr[k]["dist_reco"]
            arr[k]["bits"], arr[k]["kind"], arr[k]["cbf"] = a["bits"], a["kind"], a["cbf"]
        res = ctx.tx_eval_batch(jobs_a, arr)
        modified = {}
        for q, i in enumerate(idx):
            r, st = res[q], state[i]
            best = int(r["best"])
            modified[i] = best != -2
            if best == -2:
                continue
            s = 1 if comp else 0
            x, y, w, h = (v >> s for v in cus[i])
            if best == -1:
                st["cbf"][comp], st["tskip"][comp] = 0, 0
                st["levels"][comp] = None
                st["rec"][comp] = pred_planes[comp][y:y + h, x:x + w].copy()
                if comp == 0:
                    st["sel"] = -1
            else:
                a = per[i][best]
                st["cbf"][comp] = a["cbf"]
                st["tskip"][comp] = int(a["kind"] == api.TXE_KIND_TSKIP)
                st["levels"][comp] = a["levels"] if a["cbf"] else None
                st["rec"][comp] = a["rec"]
                if comp == 0:
                    st["sel"] = a["sel"] if a["kind"] == api.TXE_KIND_SELECT else -1
            st["cost"][comp] = int(r["cost"])
            st["reco"][comp], st["resi"][comp] = int(r["dist_reco"]), int(r["dist_resi"])
        return modified

    def root_fold(idx):
        rj = np.zeros(len(idx), api.ROOT_CBF_JOB_DTYPE)
        for q, i in enumerate(idx):
            st = state[i]
            rj[q]["lambda"] = lam
            rj[q]["dist_resi"], rj[q]["dist_reco"] = st["resi"], st["reco"]
            rj[q]["dist_zero"] = dist_zero[i]
            rj[q]["best_cu_cost"] = best_cu_cost[i]
            root = int(any(st["cbf"]))
            rj[q]["bits_non_zero"] = bits(i, 3, 0, st["cbf"], st["tskip"], st["sel"], root, st["levels"])
            rj[q]["bits_root_zero"] = bits(i, 2, 0, st["cbf"], st["tskip"], st["sel"], root, st["levels"])
            rj[q]["bits_full"] = bits(i, 4, 0, st["cbf"], st["tskip"], st["sel"], root, st["levels"])
            rj[q]["cbf"] = st["cbf"]
            rj[q]["flags"] = 1 if fast_select else 0
        return ctx.root_cbf_batch(rj)

    def apply_root(idx, rr):
        for q, i in enumerate(idx):
            st = state[i]
            st["root"] = int(rr[q]["root_cbf"])
            st["final"] = int(rr[q]["sum_dist_final"])
            if not st["root"] and any(st["cbf"]):
                for c in range(3):
                    s = 1 if c else 0
                    x, y, w, h = (v >> s for v in cus[i])
                    st["cbf"][c], st["tskip"][c] = 0, 0
                    st["levels"][c] = None
                    st["rec"][c] = pred_planes[c][y:y + h, x:x + w].copy()
                    st["reco"][c] = st["resi"][c] = int(dist_zero[i, c])
                st["sel"] = -1

    normal = [(api.TXE_KIND_NORMAL, -1), (api.TXE_KIND_TSKIP, -1)]
    select = [(api.TXE_KIND_SELECT, k) for k in range(4)]
    for comp in range(3):
        run_pass(comp, list(range(n)), normal + (select if comp == 0 and not fast_select else []),
                 True)
    rr = root_fold(list(range(n)))
    apply_root(list(range(n)), rr)
    if fast_select:
        second = [i for i in range(n) if rr[i]["second_pass"]]
        assert 0 < len(second) < n
        if second:
            mod = run_pass(0, second, select, False)
            again = [i for i in second if mod[i]]
            if again:
                apply_root(again, root_fold(again))
    # ---- compare
    stats = dict(root0=0, sel=0, tskip=0, cbf=0)
    for i in range(n):
        st, e = state[i], exp[i]
        got = (st["final"], *st["cbf"], st["root"], st["sel"], *st["tskip"],
               int(bool(merge[i]) and not any(st["cbf"])))
        want = tuple(int(v) for v in e[:10])
        assert got == want, (i, cus[i], int(inter_dir[i]), got, want)
        for c in range(3):
            assert rf.crc32_rows(st["rec"][c]) == int(e[10 + c]), (i, c, cus[i])
        stats["root0"] += not st["root"]
        stats["sel"] += st["sel"] >= 0
        stats["tskip"] += any(st["tskip"])
        stats["cbf"] += any(st["cbf"])
    return stats
    for e in envs:
        xr.xr_c1_destroy(e)
    for p in (O, R0, R1, P, Rc):
        p.destroy()


def rf_select_types():
    """tx_select idx -> (horizontal, vertical) xvcgpu_tx_type, as
    CodingUnit::SetTransformFromSelectIdx assigns them (coding_unit.cc)."""
    # inter CUs (coding_unit.cc:420-422): vertical = kInterTxMap[idx >> 1], horizontal =
    # kInterTxMap[idx & 1], kInterTxMap = {kDct8 = 3, kDst7 = 5}
    t = (3, 5)
    return {k: (t[k & 1], t[k >> 1]) for k in range(4)}
