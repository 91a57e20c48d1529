"""What a CU-state walk read back against what the reference encoder got at the same point
of the captured encode (tests/golden/rd_order_*.npz over me_calls_* / rd_calls_* /
intra_order_*): mixed into xvc_amd.cu_state's Walk / ChainedWalk by tests/rd_serial.py."""
import numpy as np

import rd_fixture as rf
from xvc_amd.cu_state import (KIND_EVAL, KIND_INTER, KIND_INTRA, KIND_MERGE_RANK, MERGE_SLOTS,
                              STATE_LIC)


class SerialChecks:
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



class ChainedChecks:
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

