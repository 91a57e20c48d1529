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
the reference reads a result (the baseline), or as one device chain per state."""
import ctypes as C

import numpy as np

import intra_fixture as ifx
import order_fixture as of
import rd_fixture as rf
from rd_replay import BL, original_planes

KIND_MERGE_RANK, KIND_EVAL, KIND_INTER, KIND_MOTION, KIND_INTRA = 0, 1, 2, 3, 4
SLOT = 64            # scratch geometry: slot k of a state at luma x = 64 * k
MAX_SLOTS = 8        # slot 0 = the prediction, 1.. = the transform alternatives

STATE_DTYPE = np.dtype([
    ("kind", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("flags", "u1"),
    ("supported", "u1"),
    ("me_first", "<i4"), ("me_count", "<i4"),
    ("bi_first", "<i4"), ("bi_count", "<i4"),
    ("aff_first", "<i4"), ("aff_uni_count", "<i4"), ("aff_bi_count", "<i4"),
    ("merge", "<i4"), ("ev", "<i4"),
    ("call_first", "<i4"), ("call_pass0", "<i4"), ("call_pass1", "<i4"),
    ("comp_count", "<i4", 3),           # pass-0 calls per component (Y.., U.., V..)
    ("copy_first", "<i4"),              # originals: [3 (slot 0)] [pass 0 calls] [pass 1 calls]
    ("cand_first", "<i4"), ("cand_count", "<i4"),
    ("final_first", "<i4"), ("final_count", "<i4"),
    ("nb_first", "<i4"), ("nb_count", "<i4"),   # LIC / intra: block copies staging the neighbours
    ("in_satd", "<i4"), ("in_first", "<i4"), ("in_count", "<i4"), ("in_reserved", "<i4"),
    ("level_first", "<i8"), ("level_count", "<i8")], align=True)
STATE_LIC = 2        # flags: the CU tries local illumination compensation
NB_WIDTH = 1024      # luma width of the neighbour staging picture


def _stream(seq):
    n = sum(len(s) for s in seq.values())
    kind = np.zeros(n, np.int8)
    idx = np.zeros(n, np.int64)
    for k, t in enumerate(of.SEQ_TABLES):
        kind[seq[t]] = k
        idx[seq[t]] = np.arange(len(seq[t]))
    return kind, idx


class Stager:
    """Sample strips of a capture (rows above, columns left, LM's luma rectangles) laid out
    in a staging picture, each copied to its place in the chain's reconstruction picture
    by one xvcgpu_copy_block.  Rows go to one-row shelves, everything taller to shelves of
    the tallest strip's height; equal content is stored once."""
    TALL = (130, 66)                    # shelf heights of the luma / chroma planes

    def __init__(self, api, width=NB_WIDTH):
        self.api, self.W = api, (width, width // 2)
        self.row = [[0, 0], [0, 0]]       # next free (x, shelf) of the one-row shelves
        self.tall = [[0, 0], [0, 0]]
        self.placed = {}                  # (cls, h, w, bytes) -> (is_row, px, shelf)
        self.data = []                    # (cls, is_row, px, shelf, array)
        self.jobs = []                    # (comp, is_row, px, shelf, dx, dy, w, h)

    def add(self, comp, dx, dy, arr):
        """arr [h, w] -> component comp at (dx, dy) of the destination; returns the job's index"""
        arr = np.ascontiguousarray(arr, np.uint16)
        h, w = arr.shape
        cls = 1 if comp else 0
        key = (cls, h, w, arr.tobytes())
        if key not in self.placed:
            is_row = h == 1
            cur = self.row[cls] if is_row else self.tall[cls]
            assert w <= self.W[cls] and h <= self.TALL[cls], (w, h)
            if cur[0] + w > self.W[cls]:
                cur[0], cur[1] = 0, cur[1] + 1
            self.placed[key] = (is_row, cur[0], cur[1])
            self.data.append((cls, is_row, cur[0], cur[1], arr))
            cur[0] += w
        is_row, px, shelf = self.placed[key]
        self.jobs.append((comp, is_row, px, shelf, dx, dy, w, h))
        return len(self.jobs) - 1

    def finish(self):
        """-> (planes of the staging picture, its height, the copy jobs)"""
        rows = [self.row[c][1] + 1 for c in range(2)]
        talls = [self.tall[c][1] + 1 for c in range(2)]
        height = max(rows[0] + self.TALL[0] * talls[0], 2 * (rows[1] + self.TALL[1] * talls[1]))
        height = (height + 63) // 64 * 64
        W = self.W[0]
        planes = [np.zeros((height, W), np.uint16), np.zeros((height // 2, W // 2), np.uint16),
                  np.zeros((height // 2, W // 2), np.uint16)]

        def sy(cls, is_row, shelf):
            return shelf if is_row else rows[cls] + self.TALL[cls] * shelf
        for cls, is_row, px, shelf, arr in self.data:
            y = sy(cls, is_row, shelf)
            for c in ((0,) if cls == 0 else (1, 2)):       # (U and V share the chroma layout)
                planes[c][y:y + arr.shape[0], px:px + arr.shape[1]] = arr
        jobs = np.zeros(len(self.jobs), self.api.COPY_BLOCK_DTYPE)
        for i, (comp, is_row, px, shelf, dx, dy, w, h) in enumerate(self.jobs):
            jobs[i] = (px, sy(1 if comp else 0, is_row, shelf), dx, dy, w, h, comp, 0)
        return planes, height, jobs


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
        sg = Stager(self.api)
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

    @staticmethod
    def _intra_strips(sg, comp, x, y, w, h, nbits, above_right, below_left, smp, off, api):
        """the reference samples of an intra prediction ([above-left] [above: w +
        above_right] [left: h + below_left], present parts only) as strips; -> samples used"""
        o = off
        if nbits & api.INTRA_HAS_ABOVE_LEFT:
            sg.add(comp, x - 1, y - 1, smp[o:o + 1].reshape(1, 1))
            o += 1
        if nbits & api.INTRA_HAS_ABOVE:
            n = w + above_right
            sg.add(comp, x, y - 1, smp[o:o + n].reshape(1, n))
            o += n
        if nbits & api.INTRA_HAS_LEFT:
            n = h + below_left
            sg.add(comp, x - 1, y, smp[o:o + n].reshape(n, 1))
            o += n
        return o - off

    def _intra_jobs(self, sg):
        """Job arrays of the intra states (kind 4): one xvcgpu_intra_block per SATD
        pre-selection, and per TransformAndReconstruct call the prediction job, the
        transform block, the quantiser's parameters and the distortion candidate."""
        api, st, io = self.api, self.states, self.intra
        S = self.state_list
        intra_states = [n for n, s_ in enumerate(S) if s_["kind"] == KIND_INTRA]
        n_satd = sum(1 for n in intra_states if S[n]["satd"] >= 0)
        call_idx = [i for n in intra_states for i in S[n]["itx"]]
        nc = len(call_idx)
        self.in_satd_jobs = np.zeros(n_satd, api.INTRA_DTYPE)
        self.in_satd_call = np.zeros(n_satd, np.int64)
        self.in_pred = np.zeros(nc, api.INTRA_DTYPE)
        self.in_tx = np.zeros(nc, api.TX_DTYPE)
        self.in_prm = np.zeros(nc, api.RDOQ_PARAMS_DTYPE)
        self.in_cand = np.zeros(nc, api.CAND_DTYPE)
        self.in_off = np.zeros(nc, np.uint32)
        self.in_ctx = np.zeros(nc, np.int32)
        self.in_weight = np.zeros(nc, np.float64)
        self.in_comp = np.zeros(nc, np.int32)
        self.in_stage = np.zeros((nc, 2), np.int32)      # per call: first / count of its block copies
        self.in_wait = np.zeros(nc, np.int32)            # a read-back behind the call (end of a mode)
        self.in_want = io["itx"][np.asarray(call_idx, np.int64)] if nc else (io["itx"][:0] if io is not None else None)
        self.n_in_levels = 0
        if io is None or not intra_states:
            self.in_contexts = np.zeros(1, api.RDOQ_CTX_DTYPE)
            return
        calls, itx = io["calls"], io["itx"]
        qps = io["qps"].view(rf.QP_DTYPE).reshape(-1)
        self.in_contexts = np.ascontiguousarray(io["contexts"]).view(api.RDOQ_CTX_DTYPE).reshape(-1)
        k_satd = k_call = 0
        for n in intra_states:
            s_ = S[n]
            r = st[n]
            r["in_satd"], r["in_first"], r["in_count"] = -1, k_call, len(s_["itx"])
            staged = {}                 # component -> the samples staged last (bytes)
            if s_["satd"] >= 0:
                c = calls[s_["satd"]]
                x, y, w, h = int(c["x"]), int(c["y"]), int(c["w"]), int(c["h"])
                first = len(sg.jobs)
                used = self._intra_strips(sg, 0, x, y, w, h, int(c["neighbors"]), int(c["above_right"]),
                                          int(c["below_left"]), io["samples"], int(c["sample_off"]), api)
                staged[0] = io["samples"][int(c["sample_off"]):int(c["sample_off"]) + used].tobytes()
                r["nb_first"], r["nb_count"] = first, len(sg.jobs) - first
                jb = self.in_satd_jobs[k_satd]
                jb["x"], jb["y"], jb["w"], jb["h"], jb["comp"] = x, y, w, h, 0
                jb["neighbors"], jb["above_right"], jb["below_left"] = c["neighbors"], c["above_right"], c["below_left"]
                self.in_satd_jobs[k_satd] = jb
                self.in_satd_call[k_satd] = s_["satd"]
                r["in_satd"] = k_satd
                k_satd += 1
            for pos_, i in enumerate(s_["itx"]):
                t = itx[i]
                comp = int(t["comp"])
                x, y, w, h = int(t["x"]), int(t["y"]), int(t["w"]), int(t["h"])
                off = int(t["sample_off"])
                nbits = int(t["neighbors"])
                n_ref = ((1 if nbits & api.INTRA_HAS_ABOVE_LEFT else 0) +
                         (w + int(t["above_right"]) if nbits & api.INTRA_HAS_ABOVE else 0) +
                         (h + int(t["below_left"]) if nbits & api.INTRA_HAS_LEFT else 0))
                lm = int(t["mode"]) == 67
                n_lm = 0
                if lm:
                    lx, ly = x << 1, y << 1
                    x0, y0 = (lx - 3 if lx > 0 else lx), (ly - 2 if ly > 0 else ly)
                    rw, rh = lx + 2 * w - x0, ly + 2 * h - y0
                    n_lm = rw * rh
                blob = io["itx_samples"][off:off + n_ref + n_lm]
                first = len(sg.jobs)
                if staged.get(comp) != blob[:n_ref].tobytes():
                    self._intra_strips(sg, comp, x, y, w, h, nbits, int(t["above_right"]),
                                       int(t["below_left"]), io["itx_samples"], off, api)
                    staged[comp] = blob[:n_ref].tobytes()
                if lm and staged.get("lm") != blob[n_ref:].tobytes():
                    sg.add(0, x0, y0, blob[n_ref:].reshape(rh, rw))
                    staged["lm"] = blob[n_ref:].tobytes()
                    staged.pop(0, None)        # (the rectangle overwrote the luma strips)
                self.in_stage[k_call] = (first, len(sg.jobs) - first)
                jb = self.in_pred[k_call]
                for f in ("x", "y", "w", "h", "comp", "mode", "neighbors", "above_right", "below_left"):
                    jb[f] = t[f]
                self.in_pred[k_call] = jb
                b = self.in_tx[k_call]
                for f in ("x", "y", "w", "h", "comp", "tx_ver", "qp", "dst4x4"):
                    b[f] = t[f]
                b["tx_hor"] = 6 if t["tx_skip"] else t["tx_hor"]
                b["intra_pic"] = api.TXF_RDOQ | (int(t["scan"]) << api.TXF_SCAN_SHIFT) | (1 if t["intra_pic"] else 0)
                self.in_tx[k_call] = b
                q = qps[int(t["qp_index"])]
                pr = self.in_prm[k_call]
                pr["lambda"], pr["rd_factor"] = q["lambda"][comp], q["rd_factor"][comp]
                pr["ctx_index"], pr["flags"] = 0, api.RDOQ_INTRA_CU
                self.in_prm[k_call] = pr
                cd = self.in_cand[k_call]
                cd["x"], cd["y"], cd["w"], cd["h"] = x, y, w, h
                cd["metric"], cd["qp"] = (7 if comp == 0 else 0), t["qp_luma"]
                self.in_cand[k_call] = cd
                self.in_off[k_call] = self.n_in_levels
                self.n_in_levels += w * h
                self.in_ctx[k_call] = t["ctx_index"]
                self.in_weight[k_call] = q["dist_weight"][comp]
                self.in_comp[k_call] = comp
                nxt = itx[s_["itx"][pos_ + 1]] if pos_ + 1 < len(s_["itx"]) else None
                self.in_wait[k_call] = int(nxt is None or int(nxt["comp"]) < comp or
                                           (comp == 0 and (int(nxt["comp"]) != 0 or int(nxt["mode"]) != int(t["mode"]))))
                k_call += 1

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
class CsTables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("orig", "refs")] + [("n_refs", C.c_int32)] + \
        [(n, C.c_void_p) for n in (
            "s_orig", "s_pred", "s_rec", "d_me", "d_me_res", "me_ref", "d_bi", "d_bi_res", "bi_ref",
            "d_aff", "d_aff_res", "aff_ref", "d_mg_inter", "d_mg_dst", "d_mg_copy", "d_mg_cands",
            "d_mg_dist", "d_ev_inter", "d_ev_dst", "d_ev_dz", "d_ev_dz_dist", "ev_weight", "ev_ctx",
            "d_contexts", "d_copy_orig", "d_call_tx", "d_call_prm", "d_call_off",
            "d_call_copy_pred", "d_call_cand", "d_levels", "d_nnz", "d_call_dist", "h_me_res",
            "h_bi_res", "h_aff_res", "h_mg_dist", "h_ev_dz_dist", "h_call_dist", "h_nnz",
            "h_levels", "rec", "nb", "d_nb_copy", "d_bi_lic",
            # intra states
            "ipred", "irec", "d_in_satd_jobs", "d_in_satd", "h_in_satd", "d_in_pred", "d_in_tx",
            "d_in_prm", "d_in_off", "d_in_cand", "d_in_contexts", "in_ctx", "in_weight", "in_comp",
            "in_stage", "in_wait", "in_off_h", "d_in_levels", "d_in_nnz", "d_in_dist", "h_in_nnz", "h_in_dist",
            "h_in_levels")]


class CsStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("states", C.c_int64), ("skipped", C.c_int64),
                ("api_calls", C.c_int64), ("round_trips", C.c_int64),
                ("seconds_by_kind", C.c_double * 5), ("states_by_kind", C.c_int64 * 5)]


class SerialRun:
    """One chain: a context (its own stream), the picture's job arrays on the device,
    scratch pictures and result arrays of its own."""

    def __init__(self, api, ctx, sp, pics, width, height):
        from xvc_amd import decoder
        self.api, self.ctx, self.sp = api, ctx, sp
        self.lib = decoder.load_host_library()
        for f in ("xvc_host_cu_state_run_serial",):
            getattr(self.lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p]
        self.orig = ctx.picture(width, height, 10)
        self.orig.upload(original_planes(width, height, sp.poc), BL)
        self.scratch = [ctx.picture(SLOT * MAX_SLOTS, 64, 10) for _ in range(3)]
        # LIC states: the chain's reconstruction picture (only the rows above / columns left
        # of such CUs are ever written: staged from `nb` in front of the state's jobs)
        self.rec = ctx.picture(width, height, 10)
        self.nb = ctx.picture(NB_WIDTH, sp.nb_height, 10)
        self.nb.upload(sp.nb_planes)
        self.refs = [pics[p] for p in sp.ref_pocs]
        self._ref_arr = (C.c_void_p * len(self.refs))(*[r.h_pic for r in self.refs])
        self._keep, self._pinned = [], []
        t = self.t = CsTables()
        t.orig, t.refs, t.n_refs = self.orig.h_pic, C.addressof(self._ref_arr), len(self.refs)
        t.s_orig, t.s_pred, t.s_rec = (p.h_pic for p in self.scratch)
        up = self._upload
        t.d_me, t.me_ref = up(sp.me_jobs), self._host(sp.me_ref)
        t.d_bi, t.bi_ref = up(sp.bi_jobs), self._host(sp.bi_ref)
        t.d_aff, t.aff_ref = up(sp.aff_jobs), self._host(sp.aff_ref)
        t.d_mg_inter, t.d_mg_dst = up(sp.mg_inter), up(sp.mg_dst)
        t.d_mg_copy, t.d_mg_cands = up(sp.mg_copy), up(sp.mg_cands)
        t.d_ev_inter, t.d_ev_dst, t.d_ev_dz = up(sp.ev_inter), up(sp.ev_dst), up(sp.ev_dz)
        t.ev_weight = self._host(np.ascontiguousarray(sp.ev_weight, np.float64))
        t.ev_ctx = self._host(np.ascontiguousarray(sp.ev_ctx, np.int32))
        t.d_contexts, t.d_copy_orig = up(sp.contexts), up(sp.copy_orig)
        t.d_call_tx, t.d_call_prm, t.d_call_off = up(sp.call_tx), up(sp.call_prm), up(sp.call_off)
        t.d_call_copy_pred, t.d_call_cand = up(sp.call_copy_pred), up(sp.call_cand)
        t.rec, t.nb = self.rec.h_pic, self.nb.h_pic
        t.d_nb_copy, t.d_bi_lic = up(sp.nb_copy), up(sp.bi_lic)
        # intra states: prediction and reconstruction at the CU's own place
        self.ipred, self.irec = ctx.picture(width, height, 10), ctx.picture(width, height, 10)
        t.ipred, t.irec = self.ipred.h_pic, self.irec.h_pic
        t.d_in_satd_jobs, t.d_in_pred, t.d_in_tx = up(sp.in_satd_jobs), up(sp.in_pred), up(sp.in_tx)
        t.d_in_prm, t.d_in_off, t.d_in_cand = up(sp.in_prm), up(sp.in_off), up(sp.in_cand)
        t.d_in_contexts = up(sp.in_contexts)
        t.in_ctx, t.in_weight, t.in_comp = self._host(sp.in_ctx), self._host(sp.in_weight), self._host(sp.in_comp)
        t.in_stage, t.in_wait = self._host(sp.in_stage), self._host(sp.in_wait)
        t.in_off_h = self._host(np.r_[sp.in_off, sp.n_in_levels].astype(np.uint32))
        res = self.res = {}
        for name, dt, n in (("me_res", api.MERES_DTYPE, len(sp.me_jobs)),
                            ("bi_res", api.MERES_DTYPE, len(sp.bi_jobs)),
                            ("aff_res", api.AFFINE_ME_RESULT_DTYPE, len(sp.aff_jobs)),
                            ("mg_dist", np.dtype("<u8"), 5 * len(sp.mg_inter)),
                            ("ev_dz_dist", np.dtype("<u8"), 3 * len(sp.ev_inter)),
                            ("call_dist", np.dtype("<u8"), len(sp.call_tx)),
                            ("nnz", np.dtype("<i4"), len(sp.call_tx)),
                            ("levels", np.dtype("<i2"), sp.n_levels),
                            ("in_satd", np.dtype("<u4"), 67 * len(sp.in_satd_jobs)),
                            ("in_nnz", np.dtype("<i4"), len(sp.in_tx)),
                            ("in_dist", np.dtype("<u8"), len(sp.in_tx)),
                            ("in_levels", np.dtype("<i2"), sp.n_in_levels)):
            nbytes = max(n, 1) * dt.itemsize
            d = ctx.alloc(nbytes)
            self._keep.append(d)
            h = self._pin(nbytes)
            C.memset(h, 0xff, nbytes)
            res[name] = np.frombuffer((C.c_char * nbytes).from_address(h), dt)[:n]
            setattr(t, "d_" + name, d.ptr)
            setattr(t, "h_" + name, h)
        ctx.sync()

    def _upload(self, arr):
        a = np.ascontiguousarray(arr).reshape(-1)
        if not len(a):
            a = np.zeros(1, a.dtype)
        b = self.ctx.buffer(a)
        self._keep.append(b)
        return b.ptr

    def _host(self, arr):
        a = np.ascontiguousarray(arr)
        self._keep.append(a)
        return a.ctypes.data

    def _pin(self, nbytes):
        p = C.c_void_p()
        self.ctx._check(self.ctx.lib.xvcgpu_host_alloc(self.ctx.h, nbytes, C.byref(p)))
        self._pinned.append(p)
        return p.value

    def run_serial(self, first=0, n=None, read_levels=True):
        st = self.sp.states
        n = len(st) - first if n is None else n
        stats = CsStats()
        rc = self.lib.xvc_host_cu_state_run_serial(self.ctx.h, C.addressof(self.t), st.ctypes.data,
                                                   first, n, int(read_levels), C.addressof(stats))
        if rc:
            raise RuntimeError("xvc_host_cu_state_run_serial: %d (%s)" % (
                rc, self.ctx.lib.xvcgpu_last_error(self.ctx.h)))
        return stats

    def check(self, first=0, n=None, levels=True, searches=True):
        """Every result the walk read back against what the reference encoder got.
        -> dict of (compared, mismatching) per table.  searches=False: the chained form
        keeps the searches' results in its own arrays (check_chained compares them)."""
        sp, res = self.sp, self.res
        sl_ = slice(first, None if n is None else first + n)
        st = sp.states[sl_]
        keep_ = st["supported"] != 0
        st = st[keep_]
        out = {}

        def rng(first_f, count_f):
            idx = [np.arange(int(a), int(a) + int(b)) for a, b in zip(st[first_f], count_f) if b]
            return np.concatenate(idx) if idx else np.zeros(0, np.int64)

        # (searches=False: only the LIC states that kept the serial form read their searches back)
        rb = 1 if searches else ((st["flags"] & STATE_LIC) != 0)
        if not searches and hasattr(sp, "folded"):
            rb = rb & ~sp.folded[sl_][keep_]
        i = rng("me_first", st["me_count"] * rb)
        w, g = sp.me_want[i], res["me_res"][i]
        out["me"] = (len(i), int(((g["fullpel_x"] != w["fullpel_x"]) | (g["fullpel_y"] != w["fullpel_y"]) |
                                  (g["mv_x"] != w["mv_x"]) | (g["mv_y"] != w["mv_y"]) |
                                  (g["subpel_dist"] != w["dist"])).sum()))
        i = rng("bi_first", st["bi_count"] * rb)
        w, g = sp.bi_want[i], res["bi_res"][i]
        out["bi"] = (len(i), int(((g["mv_x"] != w["mv"][:, 0, 0]) | (g["mv_y"] != w["mv"][:, 0, 1]) |
                                  (g["subpel_dist"] != w["dist"])).sum()))
        i = rng("aff_first", (st["aff_uni_count"] + st["aff_bi_count"]) * (1 if searches else 0))
        w, g = sp.aff_want[i], res["aff_res"][i]
        out["affine"] = (len(i), int((~((g["mv"] == w["mv"]).all(axis=(1, 2)) & (g["dist"] == w["dist"]))).sum()))
        m = st["merge"][st["kind"] == KIND_MERGE_RANK].astype(np.int64)
        if len(m):
            g = sp.mg_want[m]
            dist = res["mg_dist"].reshape(-1, 5)[m]
            cost = dist.astype(np.float64) + np.array([1, 2, 3, 4, 4.0])[None, :] * g["lambda_sqrt"][:, None]
            order = np.argsort(cost, axis=1, kind="stable")
            scost = np.take_along_axis(cost, order, 1)
            num = np.full(len(m), 4, np.int32)
            for k in range(4, -1, -1):
                num = np.where(scost[:, k] > scost[:, 0] * 1.25, k, num)
            ok = (order == g["order"]).all(1) & (scost == g["cost"]).all(1) & (num == g["num"])
            out["merge"] = (len(m), int((~ok).sum()))
        e = st["ev"][st["ev"] >= 0].astype(np.int64)
        if len(e):
            want = sp.ev_want["dist_zero"][e]
            got = res["ev_dz_dist"].reshape(-1, 3)[e]
            valid = want != np.uint64(0xffffffffffffffff)
            out["dist_zero"] = (int(valid.sum()), int(((got != want) & valid).sum()))
        sel = st[st["ev"] >= 0]
        i = rng("call_first", sel["call_pass0"] + sel["call_pass1"]) if len(sel) else np.zeros(0, np.int64)
        if len(i):
            # (rng() above iterates st; redo over the evaluation states)
            i = np.concatenate([np.arange(int(a), int(a) + int(b) + int(c)) for a, b, c in
                                zip(sel["call_first"], sel["call_pass0"], sel["call_pass1"])])
            w = sp.call_want[i]
            bad = res["nnz"][i] != w["nnz"]
            done = w["completed"] != 0
            bad |= done & (res["call_dist"][i] != w["dist"])
            if levels:
                lv = res["levels"]
                off = sp.call_off[i].astype(np.int64)
                ne = sp.call_tx["w"][i].astype(np.int64) * sp.call_tx["h"][i]
                for k in np.flatnonzero((w["nnz"] != 0) & ~bad):
                    if rf.crc32_rows(lv[off[k]:off[k] + ne[k]]) != int(w["levels_crc"][k]):
                        bad[k] = True
            out["calls"] = (len(i), int(bad.sum()))
        # intra states: every evaluated mode's SATD, every TransformAndReconstruct
        it = st[st["kind"] == KIND_INTRA]
        if len(it) and sp.intra is not None:
            io = sp.intra
            sat = res["in_satd"].reshape(-1, 67)
            done = wrong = 0
            for k in it["in_satd"][it["in_satd"] >= 0]:
                c = io["calls"][int(sp.in_satd_call[k])]
                e = io["evals"][int(c["first_eval"]):int(c["first_eval"]) + int(c["n_eval"])]
                done += len(e)
                wrong += int((sat[k, e["mode"]] != e["dist"]).sum())
            out["intra_satd"] = (done, wrong)
            i = np.concatenate([np.arange(int(a), int(a) + int(b)) for a, b in zip(it["in_first"], it["in_count"])]) \
                if it["in_count"].sum() else np.zeros(0, np.int64)
            w = sp.in_want[i]
            bad = res["in_nnz"][i] != w["nnz"]
            bad |= (w["completed"] != 0) & (res["in_dist"][i] != w["dist"])
            if levels:
                lv = res["in_levels"]
                off = sp.in_off[i].astype(np.int64)
                ne = sp.in_tx["w"][i].astype(np.int64) * sp.in_tx["h"][i]
                for k in np.flatnonzero((w["nnz"] != 0) & ~bad):
                    if rf.crc32_rows(lv[off[k]:off[k] + ne[k]].reshape(int(sp.in_tx["h"][i[k]]), -1)) != int(w["levels_crc"][k]):
                        bad[k] = True
            out["intra_calls"] = (len(i), int(bad.sum()))
            if bad.any():
                k = int(np.flatnonzero(bad)[0])
                self.first_bad_intra = (int(i[k]), tuple(w[k]), int(res["in_nnz"][i[k]]), int(res["in_dist"][i[k]]))
        return out

    def destroy(self):
        for p in self._pinned:
            self.ctx.lib.xvcgpu_host_free(self.ctx.h, p)
        for b in self._keep:
            if hasattr(b, "free"):
                b.free()
        for p in self.scratch + [self.orig, self.rec, self.nb, self.ipred, self.irec]:
            p.destroy()


# ======================================================================================
# The chained form.  Its COMPOSER - the passes (xvcgpu_cs_pass), the work arrays the device
# folds compose the searches' jobs in, the merge folds and evaluation slots, the distortion
# candidates and the op programs - is C++ product code: xvc_gpu::CuStateBuilder
# (xvc_amd/host/xvc_cu_state_builder.{h,cc}, bound by xvc_amd/cu_state_builder.py).  What is
# left here turns the captured encode into the builder's input records - what CuEncoder
# holds when it reaches a state (cu_encoder.cc:431-541, :579-642) - and, for the harness
# only, matches captured merge-candidate evaluations to the slots a merge fold fills.
# ======================================================================================
from xvc_amd import cu_state_builder as csb  # noqa: E402

R3 = csb.R3          # XVC_CS_MAX_REFS
CS_FULLPEL, CS_FORCE_L1_MVD_ZERO, CS_LIC, CS_AFFINE = 1, 2, 4, 8
CS_WHICH_UNSUPPORTED = 255
PASS_DTYPE = csb.PASS_DTYPE

RESULT_DTYPE = np.dtype([
    ("start_idx", "u1", (2, R3)), ("mvp_idx", "u1", (2, R3)), ("mv", "<i4", (2, R3, 3, 2)),
    ("dist", "<u4", (2, R3)), ("bits", "<u4", (2, R3)), ("cost", "<u4", (2, R3)),
    ("cost_list", "<u4", 2), ("cost_l1_unique", "<u4"), ("best_ref", "i1", 2),
    ("best_ref_l1_unique", "i1"), ("search_list", "u1"), ("bi_mvp_idx", "u1", R3),
    ("bi_valid", "u1"), ("bi_mv", "<i4", (R3, 3, 2)), ("bi_dist", "<u4", R3),
    ("bi_bits", "<u4", R3), ("bi_cost", "<u4", R3), ("which", "u1"), ("inter_dir", "u1"),
    ("ref_idx", "i1", 2), ("out_mvp_idx", "u1", 2), ("zero_mvd", "u1"), ("chosen", "u1"),
    ("best_cost", "<u4"), ("out_mv", "<i4", (2, 3, 2)), ("out_mvd", "<i4", (2, 2, 2))], align=True)

OP_DTYPE = csb.OP_DTYPE
(OP_MC_METRIC, OP_METRIC, OP_ME, OP_BI, OP_AFFINE, OP_COPY, OP_INTER_PRED, OP_RESIDUAL,
 OP_START_FOLD, OP_UNI_FOLD, OP_BI_FOLD, OP_FETCH, OP_SYNC, OP_EVAL_DIST, OP_MC_METRIC_REFS,
 OP_ME_REFS, OP_BI_REFS, OP_AFFINE_REFS, OP_MERGE_FOLD, OP_BI_LIC, OP_INTRA_SATD, OP_INTRA_PRED,
 OP_RESIDUAL_INTRA) = range(23)
PIC_ORIG, PIC_S_ORIG, PIC_S_PRED, PIC_S_REC, PIC_NB, PIC_REC, PIC_IPRED, PIC_IREC = 0, 1, 2, 3, 4, 5, 6, 7
BI_SLOTS = 2 * R3 * R3

MERGE_FOLD_DTYPE = csb.MERGE_FOLD_DTYPE
MERGE_RESULT_DTYPE = np.dtype([("cost", "<f8", 5), ("order", "<i4", 5), ("num", "<i4"),
                               ("reserved", "<i4", 2)], align=True)
MERGE_SLOTS = 4      # XVC_CS_MERGE_SLOTS


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


class CsEnv(C.Structure):
    _fields_ = [("orig", C.c_void_p), ("refs", C.c_void_p), ("n_refs", C.c_int32),
                ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("reserved", C.c_int32),
                ("s_orig", C.c_void_p), ("s_pred", C.c_void_p), ("s_rec", C.c_void_p),
                ("d_levels", C.c_void_p), ("d_results", C.c_void_p),
                ("rec", C.c_void_p), ("nb", C.c_void_p), ("ipred", C.c_void_p), ("irec", C.c_void_p),
                ("d_in_levels", C.c_void_p)]


class ChainedRun(SerialRun):
    """SerialRun + the arrays and the program of the chained form."""

    refs_form = True      # a SearchMotion step into all reference pictures as one launch
    no_copies = True      # originals read from the picture itself, an evaluation's alternatives
    #                       from its one prediction (xvcgpu_residual_rdoq_batch_at, the
    #                       candidates' orig_at): no block copies inside a chain
    fused_eval = True     # an evaluation's distortions priced by the launch that reconstructs
    #                       its alternatives (xvcgpu_residual_rdoq_batch_at's candidates)
    merge_fold = True     # the merge ranking folded on the device: a merge candidate's
    #                       evaluation predicts from the slot xvcgpu_cs_merge_fold filled
    lic_folds = True      # a LIC state's SearchMotion through the folds too (XVC_CS_LIC) instead
    #                       of the serial form with the capture's inputs

    def __init__(self, api, ctx, sp, pics, width, height, ref_lists):
        super().__init__(api, ctx, sp, pics, width, height)
        if not hasattr(sp, "passes"):
            compose(sp, ref_lists, self.lic_folds)
        self.lib.xvc_host_cs_run_program.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                     C.c_void_p]
        up = self._upload
        self.d = d = {}
        d["passes"], d["start_cands"] = up(sp.passes), up(sp.start_cands)
        d["aff_start_inter"], d["aff_start_dst"] = up(sp.aff_start_inter), up(sp.aff_start_dst)
        d["aff_start_cands"], d["aff_start_copy"] = up(sp.aff_start_cands), up(sp.aff_start_copy)
        d["me_work"], d["aff_work"] = up(sp.me_work), up(sp.aff_work)
        d["bi_work"] = up(np.zeros(max(sp.n_bi_slots, 1), api.BI_DTYPE))
        d["bi_lic_work"] = up(sp.bi_lic_work)
        d["ev_inter_work"] = up(sp.ev_inter_work)
        d["start_slots"], d["me_slots"] = up(sp.start_slots), up(sp.me_slots)
        d["bi_slots"], d["aff_slots"] = up(sp.bi_slots), up(sp.aff_slots)
        d["mg_fold"], d["mg_slots"] = up(sp.mg_fold), up(sp.mg_slots)
        self.cres = {}
        for name, dt, n in (("start_dist", np.dtype("<u8"), sp.n_start_dist),
                            ("me_res_c", api.MERES_DTYPE, len(sp.me_work)),
                            ("bi_res_c", api.MERES_DTYPE, sp.n_bi_slots),
                            ("aff_res_c", api.AFFINE_ME_RESULT_DTYPE, len(sp.aff_work)),
                            ("results", RESULT_DTYPE, len(sp.passes)),
                            ("ev_inter_out", api.INTER_DTYPE, 3 * len(sp.ev_inter))):
            nbytes = max(n, 1) * dt.itemsize
            if name != "ev_inter_out":
                buf = self.ctx.alloc(nbytes)
                self._keep.append(buf)
                self.ctx._check(self.ctx.lib.xvcgpu_memset(self.ctx.h, buf.ptr, 0xee, nbytes))
                d[name] = buf.ptr
            h = self._pin(nbytes)
            C.memset(h, 0xff, nbytes)
            self.cres[name] = (np.frombuffer((C.c_char * nbytes).from_address(h), dt)[:n], h)
        # Small results land in page-locked host memory the device writes directly
        # (xvcgpu_host_alloc): no copy kernel, no read-back call - they are there when
        # the chain's one wait returns.  Per evaluation state one block of distortions:
        # [3 cbf-zero (Y, U, V)] [one per TransformAndReconstruct call] (the builder's
        # ev_cands / edist_first).
        self.edist_first = sp.edist_first
        n_ed = sp.n_edist
        d["ev_cands"], d["ev_cands_copy"] = up(sp.ev_cands), up(sp.ev_cands_copy)
        d["call_pos"] = up(sp.call_pos)
        d["mg_ecands"], d["aff_start_ecands"] = up(sp.mg_ecands), up(sp.aff_start_ecands)
        self.z = {}
        for name, dt, n in (("nnz", np.dtype("<i4"), len(sp.call_tx)), ("edist", np.dtype("<u8"), n_ed),
                            ("mg_dist", np.dtype("<u8"), 5 * len(sp.mg_inter)),
                            ("mg_res", MERGE_RESULT_DTYPE, len(sp.mg_inter)),
                            ("mg_slots_out", api.INTER_DTYPE, 3 * len(sp.mg_slots))):
            nbytes = max(n, 1) * dt.itemsize
            h = self._pin(nbytes)
            C.memset(h, 0xff, nbytes)
            self.z[name] = (np.frombuffer((C.c_char * nbytes).from_address(h), dt)[:n], h)
        self.env = e = CsEnv()
        e.orig, e.refs, e.n_refs = self.t.orig, self.t.refs, self.t.n_refs
        e.pic_w, e.pic_h = width, height
        e.s_orig, e.s_pred, e.s_rec = self.t.s_orig, self.t.s_pred, self.t.s_rec
        e.d_levels, e.d_results = self.t.d_levels, d["results"]
        e.rec, e.nb = self.t.rec, self.t.nb
        e.ipred, e.irec, e.d_in_levels = self.t.ipred, self.t.irec, self.t.d_in_levels
        self.ctx.sync()

    # ---- program ---------------------------------------------------------------
    def _addrs(self):
        """xvc_csb_addrs: where the program's ops point (the serial form's tables, the
        uploads of the builder's arrays, the result arrays)."""
        if getattr(self, "_csb_addrs", None) is None:
            t, d = self.t, self.d
            a = csb.Addrs()
            for f in csb.ADDR_FIELDS:
                if f == "h_results":
                    v = self.cres["results"][1]
                elif f == "h_ev_inter_out":
                    v = self.cres["ev_inter_out"][1]
                elif f.startswith("z_"):
                    v = self.z[f[2:]][1]
                elif f in d:
                    v = d[f]
                else:
                    v = getattr(t, f)
                setattr(a, f, int(v) if v else 0)
            sp = self.sp
            self._csb_keep = k = dict(
                in_stage=np.ascontiguousarray(sp.in_stage, np.int32),
                in_ctx=np.ascontiguousarray(sp.in_ctx, np.int32),
                in_comp=np.ascontiguousarray(sp.in_comp, np.int32),
                in_weight=np.ascontiguousarray(sp.in_weight, np.float64),
                in_off=np.ascontiguousarray(sp.in_off, np.uint32),
                bi_ref=np.ascontiguousarray(sp.bi_ref, np.int8))
            i = csb.Intra()
            for name, v in k.items():
                setattr(i, name, v.ctypes.data if v.size else None)
            i.n_in, i.n_in_levels = len(sp.in_off), int(sp.n_in_levels)
            self._csb_addrs, self._csb_intra = a, i
        return self._csb_addrs, self._csb_intra

    def program(self, first, n, by_position=True, verify=True, refs_form=None, live=False):
        """Ops of the states [first, first + n) from xvc_gpu::CuStateBuilder::Program
        (xvc_cu_state_builder.cc): one chain (ending in a SYNC) per state, or per visit of a
        CU position; refs_form: a step of SearchMotion into all the CU's reference pictures
        as ONE launch; live: the chains a LIVE encoder could issue (a chain ends wherever the
        reference's control reads a cost that needs the host's entropy coder)."""
        refs_form = self.refs_form if refs_form is None else refs_form
        a, i = self._addrs()
        flags = (csb.BY_POSITION * bool(by_position) | csb.VERIFY * bool(verify) |
                 csb.REFS_FORM * bool(refs_form) | csb.LIVE * bool(live) |
                 csb.NO_COPIES * bool(self.no_copies) | csb.FUSED_EVAL * bool(self.fused_eval) |
                 csb.MERGE_FOLD * bool(self.merge_fold))
        return self.sp.builder.program(a, i, int(first), int(n), int(flags))

    def run_program(self, ops):
        stats = CsStats()
        ops = np.ascontiguousarray(ops)
        rc = self.lib.xvc_host_cs_run_program(self.ctx.h, C.addressof(self.env), ops.ctypes.data,
                                              len(ops), C.addressof(stats))
        if rc:
            raise RuntimeError("xvc_host_cs_run_program: %d (%s)" % (
                rc, self.ctx.lib.xvcgpu_last_error(self.ctx.h)))
        return stats

    def prepare(self, first=0, n=None, by_position=True, verify=True, live=False):
        """Record the program (what an encoder emits as it walks its CU tree; here a
        Python loop over the state table - keep it out of a timed region)."""
        n = len(self.sp.states) - first if n is None else n
        key = (first, n, by_position, verify, self.refs_form, live, self.merge_fold, self.no_copies,
               self.fused_eval)
        if getattr(self, "_prog_key", None) != key:
            self._prog = np.ascontiguousarray(self.program(first, n, by_position, verify, live=live))
            self._prog_key = key
        return self._prog

    def run_chained(self, first=0, n=None, by_position=True, verify=True, live=False):
        self.prepare(first, n, by_position, verify, live)
        stats = self.run_program(self._prog)
        self.collect()
        return stats

    @staticmethod
    def run_interleaved(runs, first=0, n=None, by_position=True, live=False):
        """k runs (their own contexts) driven by one thread,
        xvc_host_cs_run_programs_interleaved: a chain of one run is issued while the
        others' are executing."""
        k = len(runs)
        for r in runs:
            r.prepare(first, n, by_position, False, live)
        lib = runs[0].lib
        lib.xvc_host_cs_run_programs_interleaved.argtypes = [C.c_int, C.c_void_p, C.c_void_p,
                                                             C.c_void_p, C.c_void_p, C.c_void_p]
        lib.xvc_host_cs_run_programs_interleaved.restype = C.c_int
        ctxs = (C.c_void_p * k)(*[r.ctx.h for r in runs])
        envs = (C.c_void_p * k)(*[C.addressof(r.env) for r in runs])
        ops = (C.c_void_p * k)(*[r._prog.ctypes.data for r in runs])
        n_ops = (C.c_int64 * k)(*[len(r._prog) for r in runs])
        stats = CsStats()
        rc = lib.xvc_host_cs_run_programs_interleaved(k, ctxs, envs, ops, n_ops, C.addressof(stats))
        if rc:
            raise RuntimeError("xvc_host_cs_run_programs_interleaved: %d" % rc)
        for r in runs:
            r.collect()
        return stats

    @staticmethod
    def run_engine(runs, firsts, n, by_position=True, live=False, verify=False, streams=(), threads=1):
        """k runs through xvc_host_cs_run_programs_engine: every round the chains' next steps
        grouped by kind, one launch per kind with the chains' jobs side by side.  firsts[c]:
        the first state of run c's stretch of n states (the chains walk different parts of the
        picture: their steps do not line up).  streams: further contexts of the device.
        threads = 1: one engine, a round's groups dealt over the contexts; threads = T: T
        engines on T host threads, each with its own context and every T-th chain."""
        import threading
        import time
        k = len(runs)
        ctx = runs[0].ctx
        assert all(r.ctx is ctx for r in runs)
        progs = []
        for r, f in zip(runs, firsts):       # (recording a program is a Python loop: keep it)
            key = (f, n, by_position, verify, live)
            cache = r.__dict__.setdefault("_engine_programs", {})
            if key not in cache:
                cache.clear()
                cache[key] = np.ascontiguousarray(r.program(f, n, by_position, verify, live=live))
            progs.append(cache[key])
        lib = runs[0].lib
        lib.xvc_host_cs_run_programs_engine.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_void_p]
        lib.xvc_host_cs_run_programs_engine.restype = C.c_int
        all_ctx = [ctx] + list(streams)

        def one(ctxs, which, stats, err):
            kk = len(which)
            envs = (C.c_void_p * kk)(*[C.addressof(runs[c].env) for c in which])
            ops = (C.c_void_p * kk)(*[progs[c].ctypes.data for c in which])
            n_ops = (C.c_int64 * kk)(*[len(progs[c]) for c in which])
            hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
            rc = lib.xvc_host_cs_run_programs_engine(hs, len(ctxs), kk, envs, ops, n_ops,
                                                     C.addressof(stats))
            if rc:
                err.append("xvc_host_cs_run_programs_engine: %d (%s)" % (
                    rc, ctxs[0].lib.xvcgpu_last_error(ctxs[0].h)))

        err = []
        if threads <= 1:
            stats = CsStats()
            one(all_ctx, list(range(k)), stats, err)
        else:
            assert len(all_ctx) >= threads and k >= threads
            ctx.sync()                       # the runs' uploads, before other streams read them
            parts = [CsStats() for _ in range(threads)]
            ths = [threading.Thread(target=one, args=([all_ctx[t]], list(range(t, k, threads)), parts[t], err))
                   for t in range(threads)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            stats = CsStats()
            stats.seconds = time.perf_counter() - t0
            for pt in parts:
                for f in ("states", "round_trips", "api_calls"):
                    setattr(stats, f, getattr(stats, f) + getattr(pt, f))
        if err:
            raise RuntimeError(err[0])
        for r in runs:
            r.collect()
        return stats

    def collect(self):
        """The device-written host arrays into the result arrays check() reads."""
        st, res = self.sp.states, self.res
        res["nnz"][:] = self.z["nnz"][0]
        res["mg_dist"][:] = self.z["mg_dist"][0]
        ed = self.z["edist"][0]
        for ns in np.flatnonzero(self.edist_first >= 0):
            r = st[ns]
            a, ev, cf = int(self.edist_first[ns]), int(r["ev"]), int(r["call_first"])
            k = int(r["call_pass0"]) + int(r["call_pass1"])
            res["ev_dz_dist"][3 * ev:3 * ev + 3] = ed[a:a + 3]
            res["call_dist"][cf:cf + k] = ed[a + 3:a + 3 + k]

    def run_chained_state(self, first=0, n=None):
        return self.run_chained(first, n, by_position=False)

    def check_chained(self, first=0, n=None):
        """The folds' intermediates and choices against the capture: every priced
        candidate's final predictor, distortion, bits; SearchMotion's result; the motion
        the evaluation was run with.  -> dict of (compared, mismatching)."""
        sp = self.sp
        st = sp.states
        n = len(st) - first if n is None else n
        R = self.cres["results"][0]
        cd_all, fin = sp.tabs["cands"], sp.order["finals"]
        out = {"cands": [0, 0], "finals": [0, 0], "eval_motion": [0, 0], "merge_fold": [0, 0],
               "merge_slot_motion": [0, 0]}
        mres = self.z["mg_res"][0]
        slots_out = self.z["mg_slots_out"][0].reshape(-1, 3)
        for ns in range(first, first + n):
            s = st[ns]
            if s["supported"] and self.merge_fold and s["kind"] == KIND_MERGE_RANK:
                # the device's ranking (order, sorted costs, count) against the reference's
                m = int(s["merge"])
                g, w = mres[m], sp.mg_want[m]
                out["merge_fold"][0] += 1
                if not (np.array_equal(g["order"], w["order"]) and np.array_equal(g["cost"], w["cost"]) and
                        int(g["num"]) == int(w["num"])):
                    out["merge_fold"][1] += 1
                    self.first_bad = getattr(self, "first_bad", ("merge_fold", ns, tuple(g), tuple(w)))
            if s["supported"] and self.merge_fold and s["kind"] == KIND_EVAL and \
                    sp.ev_merge_slot[int(s["ev"])] >= 0 and \
                    first <= sp.merge_state[int(sp.ev_merge_slot[int(s["ev"])]) // MERGE_SLOTS] < first + n:
                # the motion the fold put into the slot this evaluation predicted from
                e = int(s["ev"])
                got, want = slots_out[int(sp.ev_merge_slot[e])], sp.ev_inter[e]
                out["merge_slot_motion"][0] += 1
                ok = True
                for c in range(3):
                    ok = ok and got[c]["flags"] == want[c]["flags"] and np.array_equal(got[c]["ref"], want[c]["ref"]) \
                        and got[c]["comp"] == c and got[c]["x"] == want[c]["x"] and got[c]["w"] == want[c]["w"]
                    for l in range(2):
                        if want[c]["ref"][l] >= 0:
                            ok = ok and np.array_equal(got[c]["mv"][l][:1], want[c]["mv"][l][:1])
                if not ok:
                    out["merge_slot_motion"][1] += 1
                    self.first_bad = getattr(self, "first_bad", ("merge_slot_motion", ns, got, want))
            if not s["supported"] or s["kind"] < KIND_INTER or sp.pass_count[ns] == 0:
                continue
            pf = int(sp.pass_first[ns])
            cds = cd_all[int(s["cand_first"]):int(s["cand_first"]) + int(s["cand_count"])]
            for c in cds:
                pi = pf + (1 if c["kind"] >= 2 else 0)
                r, l, k = R[pi], int(c["list"]), int(c["ref_idx"])
                out["cands"][0] += 1
                if c["kind"] in (0, 2):
                    ok = (r["dist"][l, k] == c["dist"] and r["bits"][l, k] == c["bits"] and
                          r["mvp_idx"][l, k] == c["mvp_idx"] and r["start_idx"][l, k] == c["start_mvp_idx"] and
                          np.array_equal(r["mv"][l, k][:3 if c["kind"] == 2 else 1], c["mv"][:3 if c["kind"] == 2 else 1]))
                else:
                    ok = (r["search_list"] == l and r["bi_dist"][k] == c["dist"] and r["bi_bits"][k] == c["bits"] and
                          r["bi_mvp_idx"][k] == c["mvp_idx"] and
                          np.array_equal(r["bi_mv"][k][:3 if c["kind"] == 3 else 1], c["mv"][:3 if c["kind"] == 3 else 1]))
                if not ok:
                    out["cands"][1] += 1
                    self.first_bad = getattr(self, "first_bad", ("cand", ns, tuple(c), tuple(r)))
            fs = fin[int(s["final_first"]):int(s["final_first"]) + int(s["final_count"])]
            for k, f in enumerate(fs):
                r = R[pf + k]
                out["finals"][0] += 1
                ok = r["which"] == f["which"] and r["inter_dir"] == f["inter_dir"]
                for l in range(2):
                    if f["inter_dir"] == 2 or f["inter_dir"] == l:
                        nc = 3 if (f["flags"] & 8) else 1
                        ok = ok and r["ref_idx"][l] == f["ref_idx"][l] and r["out_mvp_idx"][l] == f["mvp_idx"][l] and \
                            np.array_equal(r["out_mv"][l][:nc], f["mv"][l][:nc]) and \
                            np.array_equal(r["out_mvd"][l][:2 if nc == 3 else 1], f["mvd"][l][:2 if nc == 3 else 1])
                if not ok:
                    out["finals"][1] += 1
                    self.first_bad = getattr(self, "first_bad", ("final", ns, tuple(f), tuple(r)))
            if s["kind"] == KIND_INTER:
                e = int(s["ev"])
                got = self.cres["ev_inter_out"][0].reshape(-1, 3)[e]
                want = sp.ev_inter[e]
                out["eval_motion"][0] += 1
                ok = True
                for c in range(3):
                    ok = ok and got[c]["flags"] == want[c]["flags"] and np.array_equal(got[c]["ref"], want[c]["ref"])
                    for l in range(2):
                        if want[c]["ref"][l] >= 0:
                            nc = 3 if want[c]["flags"] & 1 else 1
                            ok = ok and np.array_equal(got[c]["mv"][l][:nc], want[c]["mv"][l][:nc])
                if not ok:
                    out["eval_motion"][1] += 1
                    self.first_bad = getattr(self, "first_bad", ("eval_motion", ns, got, want))
        return {k: tuple(v) for k, v in out.items()}


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
