"""A real picture's RD search in the ORDER the reference encoder issued it
(tests/golden/rd_order_*.npz over me_calls_* / rd_calls_*), as CU STATES:

  kind 0  merge ranking   SearchMergeCandidates (inter_search.cc:165-197)
  kind 1  evaluation      CompressAndEvalCbf (:261-365) of a given motion: a merge
                          candidate (CompressMergeCand :117-139)
  kind 2  inter mode      CompressInter (:74-98): SearchMotion [+ the affine second
                          pass] then CompressAndEvalCbf of what it chose
  kind 3  motion only     a CompressInter that returned before its evaluation
                          (whole-sample vectors with a zero difference, :94-96)
  kind 4  intra mode      CompressIntra (cu_encoder.cc:518-541): the SATD pre-selection
                          over the 67 luma modes (DetermineSlowIntraModes,
                          intra_search.cc:188-305), then PredictAndTransform of every
                          kept luma mode and every chroma mode (:61-82, :118-150) - each
                          a prediction from the reconstruction of that moment around
                          the CU + the TransformAndReconstruct alternatives
                          (tests/golden/intra_order_*.npz)

State boundaries come from the capture's global sequence numbers: a state is what the
reference computes between two points where CuEncoder (cu_encoder.cc:431-515, :598-...)
reads a cost and decides what to try next.  This module turns the fixture records
into the device job arrays of the C-ABI entry points (all states of a picture, in
issue order, uploaded once) and a table of xvc_cs_state records; the C++ layer
(xvc_amd/host/xvc_cu_state.cc) walks that table - serially with a read-back wherever
the reference reads a result (the baseline), or as one device chain per state.

The walk itself (device tables, programs, the calls into libxvchost.so) is product code:
xvc_amd/cu_state.py over xvc_amd/host/xvc_cu_state{,_builder}.cc.  Here: the capture ->
its input records (SerialPicture, builder_inputs), and SerialRun / ChainedRun = that walk
fed with the capture + the comparisons of tests/rd_checks.py."""
import ctypes as C

import numpy as np

import intra_fixture as ifx
import order_fixture as of
import rd_checks
import rd_intra_feed
import rd_fixture as rf
from rd_replay import BL, original_planes
from xvc_amd import cu_state
from xvc_amd import cu_state_builder as csb
from xvc_amd.cu_state import *  # noqa: F401,F403 (the walk's record types and constants)
from xvc_amd.cu_state import (CS_AFFINE, CS_LIC, KIND_EVAL, KIND_INTER, KIND_INTRA,  # noqa: F401
                              KIND_MERGE_RANK, KIND_MOTION, MAX_SLOTS, MERGE_SLOTS, NB_WIDTH,
                              R3, SLOT, STATE_DTYPE, STATE_LIC, CsEnv, CsStats, CsTables)

def _stream(seq):
    n = sum(len(s) for s in seq.values())
    kind = np.zeros(n, np.int8)
    idx = np.zeros(n, np.int64)
    for k, t in enumerate(of.SEQ_TABLES):
        kind[seq[t]] = k
        idx[seq[t]] = np.arange(len(seq[t]))
    return kind, idx


class SerialPicture:
    """The states of ONE picture of a clip and their device job arrays."""

    def __init__(self, api, name, poc, intra=True):
        self.api, self.name, self.poc = api, name, poc
        self.rd = rd = rf.load(name)
        self.order = o = of.load(name)
        self.me = np.load(rf.GOLDEN + "/me_calls_%s.npz" % name)["calls"]
        self.tabs = {"me": self.me, "steps": rd["steps"], "merges": rd["merges"],
                     "evals": rd["evals"], "calls": rd["calls"], "cands": o["cands"],
                     "finals": o["finals"]}
        # (intra=False: the inter states only, as rounds 4 - 5 walked them)
        self.intra = ifx.load_order(name) if intra else None
        self._group()
        self._jobs()

    # ---- state boundaries ---------------------------------------------------------
    def _group(self):
        kind, idx = _stream(self.order["seq"])
        T = of.SEQ_TABLES
        tabs = self.tabs
        ev_tab, calls = tabs["evals"], tabs["calls"]
        states = []
        pending = None

        def new_state(k, key):
            return dict(kind=k, key=key, me=[], cands=[], steps=[], finals=[], merge=-1, ev=-1,
                        calls=[])

        def flush():
            nonlocal pending
            if pending is not None:
                pending["kind"] = KIND_MOTION
                states.append(pending)
                pending = None

        cur = None           # the evaluation state collecting calls
        # the picture's intra records, merged into the order by the number of inter records
        # in front of each
        io = self.intra
        iev = []
        if io is not None:
            iev = sorted([(int(io["pos"]["calls"][i]), int(io["stamp"]["calls"][i]), 0, i)
                          for i in range(len(io["calls"]))] +
                         [(int(io["pos"]["itx"][i]), int(io["stamp"]["itx"][i]), 1, i)
                          for i in range(len(io["itx"]))])
        inext = 0
        cur_intra = None

        def intra_until(p):
            """the intra records in front of inter record p of the order"""
            nonlocal inext, cur_intra, cur
            while inext < len(iev) and iev[inext][0] <= p:
                _, _, which, i = iev[inext]
                inext += 1
                if which == 0:
                    c = io["calls"][i]
                    key = (int(c["x"]), int(c["y"]), int(c["w"]), int(c["h"]))
                    flush()
                    cur = None
                    cur_intra = new_state(KIND_INTRA, key)
                    cur_intra.update(satd=i, itx=[])
                    states.append(cur_intra)
                else:
                    t_ = io["itx"][i]
                    sh_ = 1 if t_["comp"] else 0
                    key = (int(t_["x"]) << sh_, int(t_["y"]) << sh_, int(t_["w"]) << sh_, int(t_["h"]) << sh_)
                    if cur_intra is None or cur_intra["key"] != key:
                        flush()
                        cur = None
                        cur_intra = new_state(KIND_INTRA, key)
                        cur_intra.update(satd=-1, itx=[])
                        states.append(cur_intra)
                    cur_intra["itx"].append(i)

        for p_, (k, i) in enumerate(zip(kind, idx)):
            intra_until(p_)
            t = T[k]
            r = tabs[t][i]
            if t == "calls":
                e = ev_tab[r["eval"]]
                if int(e["poc"]) != self.poc:
                    continue
                cur_intra = None
                key = (int(e["x"]), int(e["y"]), int(e["w"]), int(e["h"]))
                first = r["comp"] == 0 and r["tx_select_idx"] < 0 and not r["tx_skip"]
                if first:
                    merge = (e["flags"] & rf.FLAG_MERGE) != 0
                    if pending is not None and pending["key"] == key and not merge:
                        cur = pending
                        cur["kind"] = KIND_INTER
                        pending = None
                    else:
                        flush()
                        cur = new_state(KIND_EVAL, key)
                    cur["ev"] = int(r["eval"])
                    states.append(cur)
                assert cur is not None and cur["ev"] == int(r["eval"]), (i, cur)
                cur["calls"].append(int(i))
                continue
            if int(r["poc"]) != self.poc:
                continue
            cur_intra = None
            key = (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"]))
            if t == "evals":
                continue          # the record is created by the state's first call (next record)
            if t == "merges":
                flush()
                s = new_state(KIND_MERGE_RANK, key)
                s["merge"] = int(i)
                states.append(s)
                cur = None
                continue
            # me / cands / steps / finals: the motion search of a CompressInter
            if pending is not None and pending["key"] != key:
                flush()
            if pending is None:
                pending = new_state(KIND_MOTION, key)
                cur = None
            pending[{"me": "me", "cands": "cands", "steps": "steps", "finals": "finals"}[t]].append(int(i))
        intra_until(1 << 62)
        flush()
        # a motion group that holds TWO CompressInter calls back to back (the first one
        # returned before its evaluation): split at the finals / flag change
        out = []
        for s in states:
            out.extend(self._split_motion(s))
        self.state_list = out

    def _split_motion(self, s):
        """One state per SearchMotion chain: a CompressInter's records carry one value of
        the whole-sample / illumination flags."""
        if s["kind"] not in (KIND_INTER, KIND_MOTION) or not s["cands"]:
            return [s]
        cd = self.tabs["cands"]
        fl = [int(cd[i]["flags"]) & 3 for i in s["cands"]]
        if len(set(fl)) == 1:
            return [s]
        # leading chains without evaluation, then the one that was evaluated
        parts, seen = [], []
        for f in fl:
            if not seen or seen[-1] != f:
                seen.append(f)
        for n, f in enumerate(seen):
            last = n == len(seen) - 1
            p = dict(s)
            for t, tab in (("me", "me"), ("cands", "cands"), ("steps", "steps"), ("finals", "finals")):
                src = self.tabs[tab]
                if t == "me":
                    keep = [i for i in s[t] if (int(src[i]["fullpel_mv"]) | (2 if src[i]["use_lic"] else 0)) == f]
                else:
                    keep = [i for i in s[t] if (int(src[i]["flags"]) & 3) == f]
                p[t] = keep
            if not last:
                p["kind"], p["ev"], p["calls"] = KIND_MOTION, -1, []
            parts.append(p)
        return parts

    # ---- job arrays ---------------------------------------------------------------
    def _jobs(self):
        api, rd = self.api, self.rd
        me, steps, merges = self.me, rd["steps"], rd["merges"]
        ev_tab, calls, qps = rd["evals"], rd["calls"], rd["qps"]
        S = self.state_list
        # reference pictures of this picture
        pocs = set(int(p) for p in me["ref_poc"][me["poc"] == self.poc])
        sp = steps[steps["poc"] == self.poc]
        pocs |= set(int(p) for p in sp["ref_poc"]) | set(int(p) for p in sp["other_ref_poc"] if p >= 0)
        mp = merges[merges["poc"] == self.poc]
        pocs |= set(int(p) for p in mp["ref_poc"].reshape(-1) if p >= 0)
        ep = ev_tab[ev_tab["poc"] == self.poc]
        pocs |= set(int(p) for p in ep["ref_poc"].reshape(-1) if p >= 0)
        self.ref_pocs = sorted(pocs)
        slot = {p: i for i, p in enumerate(self.ref_pocs)}
        slot[-1] = -1

        st = np.zeros(len(S), STATE_DTYPE)
        me_idx, bi_idx, aff_idx, mg_idx = [], [], [], []
        nb_of_state = {}        # LIC state -> neighbour records (capture order)
        ev_states = []          # (state index, eval index)
        call_idx = []
        n_copy = 0
        level_pos = 0
        for n, s in enumerate(S):
            r = st[n]
            r["kind"] = s["kind"]
            r["x"], r["y"], r["w"], r["h"] = s["key"]
            r["supported"] = 1
            r["merge"], r["ev"], r["in_satd"] = -1, -1, -1
            r["me_first"], r["me_count"] = len(me_idx), len(s["me"])
            me_idx += s["me"]
            bi = [i for i in s["steps"] if steps[i]["kind"] == rf.KIND_BI]
            au = [i for i in s["steps"] if steps[i]["kind"] == rf.KIND_AFFINE_UNI]
            ab = [i for i in s["steps"] if steps[i]["kind"] == rf.KIND_AFFINE_BI]
            r["bi_first"], r["bi_count"] = len(bi_idx), len(bi)
            bi_idx += bi
            r["aff_first"], r["aff_uni_count"], r["aff_bi_count"] = len(aff_idx), len(au), len(ab)
            aff_idx += au + ab
            lic = any(steps[i]["flags"] & rf.FLAG_LIC for i in s["steps"]) or \
                any(me[i]["use_lic"] for i in s["me"])
            if s["cands"]:
                r["cand_first"], r["cand_count"] = s["cands"][0], len(s["cands"])
                assert s["cands"] == list(range(s["cands"][0], s["cands"][0] + len(s["cands"])))
                r["flags"] = int(self.tabs["cands"][s["cands"][0]]["flags"]) & 3
            if s["finals"]:
                r["final_first"], r["final_count"] = s["finals"][0], len(s["finals"])
            if s["kind"] == KIND_MERGE_RANK:
                r["merge"] = len(mg_idx)
                mg_idx.append(s["merge"])
                lic = lic or bool(merges[s["merge"]]["use_lic"].any())
            if s["ev"] >= 0:
                e = ev_tab[s["ev"]]
                lic = lic or bool(e["flags"] & rf.FLAG_LIC)
                r["ev"] = len(ev_states)
                ev_states.append((n, s["ev"]))
                cs = calls[s["calls"]]
                p0 = [i for i, c in zip(s["calls"], cs) if not (c["comp"] == 0 and c["tx_select_idx"] >= 0)]
                p1 = [i for i, c in zip(s["calls"], cs) if c["comp"] == 0 and c["tx_select_idx"] >= 0]
                assert s["calls"] == p0 + p1, s          # pass 1 = the luma selections, last
                comps = calls["comp"][p0]
                assert (np.diff(comps.astype(int)) >= 0).all()
                for c in range(3):
                    r["comp_count"][c] = int((comps == c).sum())
                r["call_first"], r["call_pass0"], r["call_pass1"] = len(call_idx), len(p0), len(p1)
                call_idx += s["calls"]
                r["copy_first"] = n_copy
                n_copy += 3 + len(s["calls"])
                sizes = [(int(e["w"]) >> (1 if calls[i]["comp"] else 0)) *
                         (int(e["h"]) >> (1 if calls[i]["comp"] else 0)) for i in s["calls"]]
                r["level_first"], r["level_count"] = level_pos, sum(sizes)
                level_pos += sum(sizes)
            if lic:
                # local illumination compensation: the model reads the reconstruction of
                # THAT MOMENT above / left of the CU (the capture's neighbour records)
                r["flags"] = int(r["flags"]) | STATE_LIC
                ks = [int(steps[i]["nb_index"]) for i in s["steps"] if steps[i]["flags"] & rf.FLAG_LIC]
                if s["kind"] == KIND_MERGE_RANK and merges[s["merge"]]["use_lic"].any():
                    ks.append(int(merges[s["merge"]]["nb_index"]))
                if s["ev"] >= 0 and (ev_tab[s["ev"]]["flags"] & rf.FLAG_LIC):
                    ks.append(int(ev_tab[s["ev"]]["nb_index"]))
                assert all(k >= 0 for k in ks), (n, ks)
                nb_of_state[n] = list(dict.fromkeys(ks))
        self.states = st
        self.nb_of_state = nb_of_state
        self._stage_neighbours(nb_of_state)
        self.n_levels = level_pos

        i_ = np.asarray
        # -- uni-directional searches
        m = me[i_(me_idx, np.int64)] if me_idx else me[:0]
        self.me_jobs = np.zeros(len(m), api.ME_DTYPE)
        for k in ("x", "y", "w", "h", "depth_nonzero", "mvp_x", "mvp_y", "prev_x", "prev_y",
                  "lambda16", "search_range"):
            self.me_jobs[k] = m[k]
        self.me_jobs["fullpel_mv"] = m["fullpel_mv"] | np.where(m["use_lic"] != 0, 2, 0)
        self.me_ref = np.array([slot[int(p)] for p in m["ref_poc"]], np.int8)
        self.me_want = m
        # -- bi-prediction refinement steps
        b = steps[i_(bi_idx, np.int64)] if bi_idx else steps[:0]
        self.bi_jobs = np.zeros(len(b), api.BI_DTYPE)
        blk = self.bi_jobs["blk"]
        for k in ("x", "y", "w", "h", "lambda16"):
            blk[k] = b[k]
        blk["fullpel_mv"] = (b["flags"] & rf.FLAG_FULLPEL) != 0
        ar = np.arange(len(b))
        stt = b["start_mvp_idx"].astype(np.int64)
        blk["mvp_x"], blk["mvp_y"] = b["mvp"][ar, stt, 0, 0], b["mvp"][ar, stt, 0, 1]
        blk["search_range"] = 4
        self.bi_jobs["blk"] = blk
        self.bi_jobs["other_mv_x"], self.bi_jobs["other_mv_y"] = b["other_mv"][:, 0, 0], b["other_mv"][:, 0, 1]
        self.bi_jobs["boot_mv_x"], self.bi_jobs["boot_mv_y"] = b["boot"][:, 0, 0], b["boot"][:, 0, 1]
        self.bi_ref = np.array([[slot[int(p)], slot[int(q)]] for p, q in
                                zip(b["ref_poc"], b["other_ref_poc"])], np.int8).reshape(-1, 2)
        self.bi_want = b
        self.bi_lic = self._lic_blocks(b["x"], b["y"], b["w"], b["h"],
                                       np.where((b["flags"] & rf.FLAG_LIC) != 0, b["nb_index"], -1))
        # -- affine searches
        a = steps[i_(aff_idx, np.int64)] if aff_idx else steps[:0]
        self.aff_jobs = np.zeros(len(a), api.AFFINE_ME_DTYPE)
        for k in ("x", "y", "w", "h", "lambda16"):
            self.aff_jobs[k] = a[k]
        self.aff_jobs["flags"] = (np.where((a["flags"] & rf.FLAG_HAS_BOOT) != 0, api.AFFINE_ME_HAS_BOOTSTRAP, 0) |
                                  np.where(a["kind"] == rf.KIND_AFFINE_BI, api.AFFINE_ME_BIPRED, 0))
        ar = np.arange(len(a))
        self.aff_jobs["mvp"] = a["mvp"][ar, a["start_mvp_idx"].astype(np.int64)]
        self.aff_jobs["bootstrap"] = a["boot"]
        self.aff_jobs["other_mv"] = a["other_mv"]
        self.aff_ref = np.array([[slot[int(p)], slot[int(q)] if q >= 0 else slot[int(p)]]
                                 for p, q in zip(a["ref_poc"], a["other_ref_poc"])], np.int8).reshape(-1, 2)
        self.aff_want = a
        # -- merge rankings: five luma predictions + SATD per call
        g = merges[i_(mg_idx, np.int64)] if mg_idx else merges[:0]
        n = len(g)
        self.mg_want = g
        self.mg_inter = np.zeros((n, 5), api.INTER_DTYPE)
        j = self.mg_inter
        j["x"], j["y"], j["w"], j["h"] = g["x"][:, None], g["y"][:, None], g["w"][:, None], g["h"][:, None]
        for l in range(2):
            used = (g["inter_dir"] == 2) | (g["inter_dir"] == l)
            j["ref"][:, :, l] = np.where(used, np.vectorize(lambda p: slot.get(int(p), -1))(g["ref_poc"][:, :, l]), -1)
        j["mv"][:, :, :, 0, :] = g["mv"]
        self._lic_fields(j, g["use_lic"] != 0, np.repeat(g["nb_index"][:, None], 5, 1))
        self.mg_dst = np.zeros((n, 5), api.POS_DTYPE)
        self.mg_dst["x"] = SLOT * np.arange(5)[None, :]
        self.mg_copy = np.zeros((n, 5), api.COPY_BLOCK_DTYPE)
        c = self.mg_copy
        c["sx"], c["sy"] = g["x"][:, None], g["y"][:, None]
        c["dx"] = SLOT * np.arange(5)[None, :]
        c["w"], c["h"] = g["w"][:, None], g["h"][:, None]
        self.mg_cands = np.zeros((n, 5), api.CAND_DTYPE)
        k = self.mg_cands
        k["x"] = SLOT * np.arange(5)[None, :]
        k["w"], k["h"], k["metric"] = g["w"][:, None], g["h"][:, None], 1
        # -- evaluations
        ne = len(ev_states)
        e = ev_tab[i_([x[1] for x in ev_states], np.int64)] if ne else ev_tab[:0]
        self.ev_want = e
        self.ev_state = np.array([x[0] for x in ev_states], np.int64)
        self.ev_inter = np.zeros((ne, 3), api.INTER_DTYPE)
        j = self.ev_inter
        for c in range(3):
            jc = j[:, c]
            jc["x"], jc["y"], jc["w"], jc["h"], jc["comp"] = e["x"], e["y"], e["w"], e["h"], c
            jc["flags"] = np.where((e["flags"] & rf.FLAG_AFFINE) != 0, api.INTER_AFFINE, 0)
            for l in range(2):
                used = (e["inter_dir"] == 2) | (e["inter_dir"] == l)
                jc["ref"][:, l] = np.where(used, [slot.get(int(p), -1) for p in e["ref_poc"][:, l]], -1)
            jc["mv"] = e["mv"]
            j[:, c] = jc
        self._lic_fields(j, np.repeat(((e["flags"] & rf.FLAG_LIC) != 0)[:, None], 3, 1),
                         np.repeat(e["nb_index"][:, None], 3, 1))
        self.ev_dst = np.zeros((ne, 3), api.POS_DTYPE)      # slot 0
        self.ev_weight = qps["dist_weight"][e["qp_index"]] if ne else np.zeros((0, 3))
        self.ev_ctx = e["ctx_index"].astype(np.int32)
        # dist_zero: prediction (slot 0) against the original (slot 0), per component
        self.ev_dz = np.zeros((ne, 3), api.CAND_DTYPE)
        for c in range(3):
            d = self.ev_dz[:, c]
            d["w"], d["h"] = e["w"] >> (1 if c else 0), e["h"] >> (1 if c else 0)
            d["metric"], d["qp"] = (7 if c == 0 else 0), e["qp"][:, 0]
            self.ev_dz[:, c] = d
        # -- the transform calls
        cl = calls[i_(call_idx, np.int64)] if call_idx else calls[:0]
        nc = len(cl)
        self.call_want = cl
        # which evaluation state / slot each call belongs to
        call_ev = np.zeros(nc, np.int64)
        call_slot = np.zeros(nc, np.int64)
        for n in np.flatnonzero(st["ev"] >= 0):
            r = st[n]
            f, p0, p1 = int(r["call_first"]), int(r["call_pass0"]), int(r["call_pass1"])
            call_ev[f:f + p0 + p1] = r["ev"]
            pos = f
            for c in range(3):
                k = int(r["comp_count"][c])
                call_slot[pos:pos + k] = 1 + np.arange(k)
                pos += k
            call_slot[pos:pos + p1] = 1 + int(r["comp_count"][0]) + np.arange(p1)
        assert nc == 0 or call_slot.max() < MAX_SLOTS
        ce = e[call_ev]
        self.call_ev = call_ev
        sh = (cl["comp"] != 0).astype(np.int64)
        self.call_tx = np.zeros(nc, api.TX_DTYPE)
        t = self.call_tx
        t["x"], t["y"] = (SLOT * call_slot) >> sh, 0
        t["w"], t["h"] = ce["w"] >> sh, ce["h"] >> sh
        t["comp"] = cl["comp"]
        t["tx_hor"] = np.where(cl["tx_skip"] != 0, 6, cl["tx_hor"])
        t["tx_ver"] = cl["tx_ver"]
        t["qp"] = ce["qp"][np.arange(nc), cl["comp"]]
        t["intra_pic"] = api.TXF_RDOQ | (cl["scan"].astype(np.int64) << api.TXF_SCAN_SHIFT)
        self.call_prm = np.zeros(nc, api.RDOQ_PARAMS_DTYPE)
        q = qps[ce["qp_index"]]
        self.call_prm["lambda"] = q["lambda"][np.arange(nc), cl["comp"]]
        self.call_prm["rd_factor"] = q["rd_factor"][np.arange(nc), cl["comp"]]
        self.call_prm["ctx_index"] = 0                       # one snapshot per state
        n_el = t["w"].astype(np.int64) * t["h"]
        self.call_off = np.r_[0, np.cumsum(n_el)[:-1]].astype(np.uint32) if nc else np.zeros(0, np.uint32)
        assert self.n_levels == int(n_el.sum())
        self.call_cand = np.zeros(nc, api.CAND_DTYPE)
        d = self.call_cand
        d["x"], d["y"], d["w"], d["h"] = t["x"], t["y"], t["w"], t["h"]
        d["metric"] = np.where(cl["comp"] == 0, 7, 0)
        d["qp"] = ce["qp"][:, 0]
        # prediction: slot 0 -> the call's slot (same picture)
        self.call_copy_pred = np.zeros(nc, api.COPY_BLOCK_DTYPE)
        p = self.call_copy_pred
        p["dx"], p["w"], p["h"], p["comp"] = t["x"], t["w"], t["h"], cl["comp"]
        # originals: per state [slot 0: Y U V] [pass-0 calls] [pass-1 calls]
        self.copy_orig = np.zeros(n_copy, api.COPY_BLOCK_DTYPE)
        for n in np.flatnonzero(st["ev"] >= 0):
            r = st[n]
            ev = e[int(r["ev"])]
            f = int(r["copy_first"])
            for c in range(3):
                s_ = 1 if c else 0
                self.copy_orig[f + c] = (int(ev["x"]) >> s_, int(ev["y"]) >> s_, 0, 0,
                                         int(ev["w"]) >> s_, int(ev["h"]) >> s_, c, 0)
            cf, k = int(r["call_first"]), int(r["call_pass0"]) + int(r["call_pass1"])
            o = self.copy_orig[f + 3:f + 3 + k]
            sh_ = sh[cf:cf + k]
            o["sx"], o["sy"] = ev["x"] >> sh_, ev["y"] >> sh_
            o["dx"], o["w"], o["h"], o["comp"] = t["x"][cf:cf + k], t["w"][cf:cf + k], t["h"][cf:cf + k], cl["comp"][cf:cf + k]
            self.copy_orig[f + 3:f + 3 + k] = o
        self.contexts = np.ascontiguousarray(rd["contexts"]).view(api.RDOQ_CTX_DTYPE).reshape(-1)

    # ---- local illumination compensation ----------------------------------------------
    def _nb_table(self):
        nb = self.rd["neighbours"]
        return nb if len(nb) else np.zeros(1, nb.dtype)      # (a clip without LIC CUs)

    def _lic_fields(self, jobs, lic, nb_index):
        """XVC_INTER_LIC + the neighbour fields of xvcgpu_inter_block for the jobs with lic
        (arrays of the jobs' shape)."""
        nb = self._nb_table()
        has = lic & (nb_index >= 0)
        k = np.where(has, nb_index, 0)
        jobs["flags"] = jobs["flags"] | np.where(lic, self.api.INTER_LIC, 0).astype(np.uint8)
        jobs["neighbors"] = np.where(has, nb["has_above"][k] * 1 + nb["has_left"][k] * 2, 0)
        for f in ("above_x", "above_y", "left_x", "left_y"):
            jobs[f] = np.where(has, nb[f][k], 0)

    def _lic_blocks(self, x, y, w, h, nb_index):
        """xvcgpu_mc_lic_block per job (xvcgpu_bipred_search_lic's d_neighbours)."""
        nb = self._nb_table()
        q = np.zeros(len(x), self.api.LIC_DTYPE)
        has = nb_index >= 0
        k = np.where(has, nb_index, 0)
        q["x"], q["y"], q["w"], q["h"] = x, y, w, h
        q["neighbors"] = np.where(has, nb["has_above"][k] * 1 + nb["has_left"][k] * 2, 0)
        for f in ("above_x", "above_y", "left_x", "left_y"):
            q[f] = np.where(has, nb[f][k], 0)
        return q

    def _stage_neighbours(self, nb_of_state):
        """The rows above / columns left of the LIC states' CUs as the capture holds them, and
        the reference samples of the intra states' predictions, laid out in a staging picture
        (Stager): every strip one xvcgpu_copy_block into the chain's reconstruction picture at
        the CU's place; per state (and per intra call) the range of its block copies."""
        st = self.states
        nb, smp = self.rd["neighbours"], self.rd["nb_samples"]
        sg = rd_intra_feed.Stager(self.api)
        done = {}                       # LIC neighbour record -> its jobs' indices
        ranges = {}
        for n in sorted(nb_of_state):
            first = len(sg.jobs)
            for k in nb_of_state[n]:
                r = nb[k]
                off = int(r["sample_off"])
                for c in range(3):
                    cls = 1 if c else 0
                    x, y = int(r["x"]) >> cls, int(r["y"]) >> cls
                    w, h = int(r["w"]) >> cls, int(r["h"]) >> cls
                    if r["has_above"]:
                        sg.add(c, x, y - 1, smp[off:off + w].reshape(1, w))
                        off += w
                    if r["has_left"]:
                        sg.add(c, x - 1, y, smp[off:off + h].reshape(h, 1))
                        off += h
                assert off - int(r["sample_off"]) == int(r["sample_count"]), (k, off, r)
            st["nb_first"][n], st["nb_count"][n] = first, len(sg.jobs) - first
        self._intra_jobs(sg)
        self.nb_planes, self.nb_height, self.nb_copy = sg.finish()

    def _intra_jobs(self, sg):
        rd_intra_feed.intra_jobs(self, sg)

    def position_start(self, i):
        """The first state at or behind i that opens a visit of a CU position (a stretch of
        states cut anywhere else would start inside a chain: its first states read what
        states in front of the cut computed)."""
        st = self.states
        while 0 < i < len(st) and tuple(st[i][["x", "y", "w", "h"]]) == tuple(st[i - 1][["x", "y", "w", "h"]]):
            i += 1
        return min(i, len(st) - 1)

    def representative_start(self, n, step=97):
        """The first state (one that opens a visit of a CU position) of the stretch of n
        states whose mix of state kinds - and of intra transform calls - is closest to the
        whole picture's."""
        st = self.states
        N = len(st)
        if n >= N:
            return 0
        feat = np.zeros((N, 6))
        for k in range(5):
            feat[:, k] = st["kind"] == k
        feat[:, 5] = st["in_count"] / 8.0          # (an intra state is its calls)
        cs = np.vstack([np.zeros(6), np.cumsum(feat, 0)])
        want = cs[-1] / N
        best, best_d = 0, None
        for f in range(0, N - n, step):
            a = self.position_start(f)
            if a + n > N:
                break
            d = float(np.abs((cs[a + n] - cs[a]) / n - want).sum())
            if best_d is None or d < best_d:
                best, best_d = a, d
        return best

    def summary(self, first=0, n=None):
        st = self.states[first:None if n is None else first + n]
        return {"states": len(st), "merge_rank": int((st["kind"] == 0).sum()),
                "eval": int((st["kind"] == 1).sum()), "inter": int((st["kind"] == 2).sum()),
                "motion_only": int((st["kind"] == 3).sum()), "intra": int((st["kind"] == KIND_INTRA).sum()),
                "intra_calls": int(st["in_count"].sum()),
                "unsupported": int((st["supported"] == 0).sum()),
                "me": len(self.me_jobs), "bi": len(self.bi_jobs), "affine": len(self.aff_jobs),
                "calls": len(self.call_tx)}


# ---- running the walk (needs the device) ----------------------------------------------

# ======================================================================================
# The chained form's COMPOSER is C++ product code: xvc_gpu::CuStateBuilder
# (xvc_amd/host/xvc_cu_state_builder.{h,cc}, bound by xvc_amd/cu_state_builder.py).  Here the
# captured encode becomes the builder's input records - what CuEncoder holds when it reaches
# a state (cu_encoder.cc:431-541, :579-642) - and, for the harness only, captured
# merge-candidate evaluations are matched to the slots a merge fold fills.
# ======================================================================================


def match_merge_slots(sp):
    """For the harness only: which evaluation slot of a merge fold each captured
    merge-candidate evaluation corresponds to (its motion = a ranked candidate's) - the chain
    then predicts from the SLOT the fold filled, not from the capture's job."""
    st = sp.states
    ev_slot = np.full(len(sp.ev_inter), -1, np.int64)
    cur, used = None, set()
    for n in range(len(st)):
        s = st[n]
        key = (int(s["x"]), int(s["y"]), int(s["w"]), int(s["h"]))
        kind = int(s["kind"])
        if kind == KIND_MERGE_RANK and s["supported"]:
            cur, used = (int(s["merge"]), key), set()
            continue
        if kind != KIND_EVAL or cur is None or cur[1] != key or not s["supported"]:
            if kind != KIND_EVAL:
                cur = None
            continue
        m, e = cur[0], int(s["ev"])
        if not (int(sp.ev_want["flags"][e]) & rf.FLAG_MERGE):
            continue
        want = sp.ev_inter[e, 0]
        for r in range(int(sp.mg_want["num"][m])):
            cand = sp.mg_inter[m, int(sp.mg_want["order"][m][r])]
            if r not in used and int(want["flags"]) == int(cand["flags"]) and \
                    (want["ref"] == cand["ref"]).all() and \
                    all((want["mv"][l][0] == cand["mv"][l][0]).all() for l in range(2) if want["ref"][l] >= 0):
                ev_slot[e] = MERGE_SLOTS * m + r
                used.add(r)
                break
    return ev_slot


def builder_inputs(sp, ref_lists, lic_folds=True):
    """The captured picture as xvc_csb_picture's records (xvc_cu_state_builder.h): per inter /
    motion state the (list, picture) entries SearchRefIdx visits with their AMVP predictors,
    lambda, the whole-sample flag and the inter contexts; per merge ranking its CU and
    sqrt(lambda); per evaluation its position, cbf-zero candidates and weights."""
    st = sp.states
    cd_all = sp.tabs["cands"]
    nbt = sp._nb_table()
    nb = np.zeros(len(nbt), csb.NEIGHBOURS_DTYPE)
    for f in ("has_above", "has_left", "above_x", "above_y", "left_x", "left_y"):
        nb[f] = nbt[f]
    motions, entries = [], []
    for n in np.flatnonzero((st["kind"] == KIND_INTER) | (st["kind"] == KIND_MOTION)):
        s = st[n]
        cds = cd_all[int(s["cand_first"]):int(s["cand_first"]) + int(s["cand_count"])]
        # (the folds run the default SearchMotion: xvcgpu_types.h; the captured encodes are such)
        assert not cds["force_mvd_zero_other"].any(), "forced zero L1 mvd: XVC_CS_WHICH_UNSUPPORTED"
        for kind_bi in (1, 3):
            assert (cds["kind"] == kind_bi).sum() <= max(len(ref_lists[0]), len(ref_lists[1])), \
                "refinement iterations > 1: XVC_CS_WHICH_UNSUPPORTED"
        m = np.zeros((), csb.MOTION_DTYPE)
        m["state"] = n
        m["nb"] = sp.nb_of_state[n][0] if sp.nb_of_state.get(n) else -1
        for name, kind in (("plain", 0), ("affine", 2)):
            cu = cds[cds["kind"] == kind]
            m[name]["first"], m[name]["n"] = len(entries), len(cu)
            if len(cu):
                m[name]["lambda16"] = cu[0]["lambda16"]
                m[name]["fullpel"] = int(cu[0]["flags"]) & 1
                m[name]["ictx"] = sp.order["ictx"][int(cu[0]["ictx_index"])]
            for c in cu:
                e = np.zeros((), csb.REF_ENTRY_DTYPE)
                e["list"], e["ref_idx"], e["reused"], e["mvp"] = c["list"], c["ref_idx"], c["reused"], c["mvp"]
                entries.append(e)
        motions.append(m)
    w = sp.mg_want
    mg = np.zeros(len(sp.mg_inter), csb.MERGE_DTYPE)
    mg["lambda_sqrt"], mg["x"], mg["y"], mg["w"], mg["h"] = w["lambda_sqrt"], w["x"], w["y"], w["w"], w["h"]
    mg["any_lic"] = (w["use_lic"] != 0).any(1) if len(w) else 0
    mg["nb"], mg["state"] = w["nb_index"], -1
    for n in np.flatnonzero(st["kind"] == KIND_MERGE_RANK):
        mg["state"][int(st["merge"][n])] = n
    ev = np.zeros(len(sp.ev_inter), csb.EVAL_DTYPE)
    ev["x"], ev["y"] = sp.ev_want["x"], sp.ev_want["y"]
    ev["dz"], ev["weight"] = sp.ev_dz, sp.ev_weight
    ev["merge_slot"] = sp.ev_merge_slot
    return dict(states=st, ref_lists=ref_lists, slot_pocs=np.asarray(sp.ref_pocs, np.int32),
                lic_folds=lic_folds,
                motions=np.array(motions, csb.MOTION_DTYPE) if motions else np.zeros(0, csb.MOTION_DTYPE),
                entries=np.array(entries, csb.REF_ENTRY_DTYPE) if entries else np.zeros(0, csb.REF_ENTRY_DTYPE),
                nb=nb, me_jobs=sp.me_jobs, me_ref=np.asarray(sp.me_ref, np.int8), aff_jobs=sp.aff_jobs,
                aff_ref=np.asarray(sp.aff_ref, np.int8).reshape(-1), ev_inter=sp.ev_inter.reshape(-1),
                merges=mg, evals=ev, ev_ctx=np.asarray(sp.ev_ctx, np.int32), call_cand=sp.call_cand,
                call_comp=np.asarray(sp.call_tx["comp"], np.uint8),
                call_ev=np.asarray(sp.call_ev, np.int32), mg_cands=sp.mg_cands.reshape(-1))


def compose(sp, ref_lists, lic_folds=True):
    """Runs xvc_gpu::CuStateBuilder on the picture and hangs its arrays on sp (passes,
    pass_first / pass_count per state, work arrays, slots, merge folds, candidates)."""
    sp.ev_merge_slot = match_merge_slots(sp)
    sp.builder = b = csb.Builder(builder_inputs(sp, ref_lists, lic_folds))
    for name, _ in csb.ARRAYS:
        setattr(sp, name, getattr(b, name))
    sp.folded = sp.folded.astype(bool)
    sp.bi_slots, sp.aff_slots = sp.bi_slots.reshape(-1, 2), sp.aff_slots.reshape(-1, 2)
    sp.ev_inter_work, sp.mg_slots = sp.ev_inter_work.reshape(-1, 3), sp.mg_slots.reshape(-1, 3)
    sp.n_start_dist, sp.n_bi_slots, sp.n_edist = b.n_start_dist, b.n_bi_slots, b.n_edist
    return b


class SerialRun(rd_checks.SerialChecks, cu_state.Walk):
    """The serial walk of a captured picture + check()."""

    def __init__(self, api, ctx, sp, pics, width, height):
        super().__init__(api, ctx, sp, pics, width, height,
                         original_planes(width, height, sp.poc), BL)


class ChainedRun(rd_checks.SerialChecks, rd_checks.ChainedChecks, cu_state.ChainedWalk):
    """The chained / interleaved / engine walk of a captured picture + the checks."""

    def __init__(self, api, ctx, sp, pics, width, height, ref_lists):
        if not hasattr(sp, "passes"):
            compose(sp, ref_lists, self.lic_folds)
        super().__init__(api, ctx, sp, pics, width, height,
                         original_planes(width, height, sp.poc), BL)


def ref_lists_of(name, poc):
    """([picture per ref_idx of list 0], [... of list 1]) of the clip's picture `poc`
    (stream fixture: the reference encoder's own lists)."""
    import stream_fixture as sf
    fx = sf.StreamFixture(name)
    for i in range(fx.n):
        info = fx.info[i]
        if int(info["poc"]) == poc:
            return tuple([int(info["ref_poc"][l][k]) for k in range(int(info["num_ref"][l]))]
                         for l in range(2))
    raise KeyError(poc)
