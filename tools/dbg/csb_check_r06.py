"""Round 6, one-off: held xvc_gpu::CuStateBuilder (C++) against the Python composer of rounds 4 - 5
(tests/rd_serial.py build_passes / build_merge_folds / ChainedRun.program as of commit c415688) on
tiny, c0 (POC 2, 4) and c1: every array and 8 program variants x 2 ranges byte-equal.  The Python
composer was removed after that run; this script is kept as the record of what was compared (it
needs that commit's tests/rd_serial.py to run)."""
import sys, os, time
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, ctypes as C
import rd_serial as rs
from xvc_amd import api, cu_state_builder as csb

def builder_inputs(sp, ref_lists, lic_folds=True):
    st = sp.states
    cd_all = sp.tabs["cands"]
    nbt = sp._nb_table()
    nb = np.zeros(len(nbt), csb.NEIGHBOURS_DTYPE)
    for f in ("has_above","has_left","above_x","above_y","left_x","left_y"): nb[f] = nbt[f]
    motions, entries = [], []
    for n in np.flatnonzero(((st["kind"] == rs.KIND_INTER) | (st["kind"] == rs.KIND_MOTION))):
        s = st[n]
        cds = cd_all[int(s["cand_first"]):int(s["cand_first"]) + int(s["cand_count"])]
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
    mg = np.zeros(len(sp.mg_inter), csb.MERGE_DTYPE)
    w = sp.mg_want
    mg["lambda_sqrt"], mg["x"], mg["y"], mg["w"], mg["h"] = w["lambda_sqrt"], w["x"], w["y"], w["w"], w["h"]
    mg["any_lic"] = (w["use_lic"] != 0).any(1) if len(w) else 0
    mg["nb"] = w["nb_index"]
    mg["state"] = -1
    for n in np.flatnonzero(st["kind"] == rs.KIND_MERGE_RANK):
        mg["state"][int(st["merge"][n])] = n
    ev = np.zeros(len(sp.ev_inter), csb.EVAL_DTYPE)
    ev["x"], ev["y"] = sp.ev_want["x"], sp.ev_want["y"]
    ev["dz"], ev["weight"] = sp.ev_dz, sp.ev_weight
    ev["merge_slot"] = sp.ev_merge_slot if hasattr(sp, "ev_merge_slot") else -1
    return dict(states=st, ref_lists=ref_lists, slot_pocs=np.asarray(sp.ref_pocs, np.int32), lic_folds=lic_folds,
                motions=np.array(motions, csb.MOTION_DTYPE) if motions else np.zeros(0, csb.MOTION_DTYPE),
                entries=np.array(entries, csb.REF_ENTRY_DTYPE) if entries else np.zeros(0, csb.REF_ENTRY_DTYPE),
                nb=nb, me_jobs=sp.me_jobs, me_ref=np.asarray(sp.me_ref, np.int8), aff_jobs=sp.aff_jobs,
                aff_ref=np.asarray(sp.aff_ref, np.int8).reshape(-1), ev_inter=sp.ev_inter.reshape(-1),
                merges=mg, evals=ev, ev_ctx=np.asarray(sp.ev_ctx, np.int32), call_cand=sp.call_cand,
                call_comp=np.asarray(sp.call_tx["comp"], np.uint8), call_ev=np.asarray(sp.call_ev, np.int32),
                mg_cands=sp.mg_cands.reshape(-1))

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
poc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sp = rs.SerialPicture(api, name, poc)
rl = rs.ref_lists_of(name, poc)
t=time.time(); rs.build_passes(sp, rl, True); rs.build_merge_folds(sp); print("python build", time.time()-t)
t=time.time(); inp = builder_inputs(sp, rl, True); print("inputs", time.time()-t)
t=time.time(); b = csb.Builder(inp); print("c++ build", time.time()-t)
pairs = dict(passes=sp.passes, pass_first=sp.pass_first, pass_count=sp.pass_count, folded=sp.folded.astype(np.uint8),
             start_cands=sp.start_cands, start_slots=sp.start_slots, aff_start_inter=sp.aff_start_inter,
             aff_start_dst=sp.aff_start_dst, aff_start_cands=sp.aff_start_cands, aff_start_copy=sp.aff_start_copy,
             me_work=sp.me_work, bi_lic_work=sp.bi_lic_work, aff_work=sp.aff_work, aff_work_src=sp.aff_work_src,
             me_slots=sp.me_slots, bi_slots=sp.bi_slots.reshape(-1), aff_slots=sp.aff_slots.reshape(-1),
             ev_inter_work=sp.ev_inter_work.reshape(-1), mg_fold=sp.mg_fold, mg_slots=sp.mg_slots.reshape(-1),
             merge_state=sp.merge_state)
bad = 0
for k, want in pairs.items():
    got = getattr(b, k)
    w = np.ascontiguousarray(want)
    ok = got.tobytes() == w.tobytes()
    if not ok:
        bad += 1
        print("MISMATCH", k, got.shape, w.shape, got.dtype.itemsize, w.dtype.itemsize)
        if got.nbytes == w.nbytes:
            g8, w8 = np.frombuffer(got.tobytes(), np.uint8), np.frombuffer(w.tobytes(), np.uint8)
            d = np.flatnonzero(g8 != w8); print("   first diff byte", d[:10], "of", len(d), "item", d[0] // got.dtype.itemsize, "off", d[0] % got.dtype.itemsize)
print("n_start_dist", b.n_start_dist, sp.n_start_dist, "n_bi_slots", b.n_bi_slots, sp.n_bi_slots)
print("arrays bad:", bad)

# ---- evaluation candidates (ChainedRun.__init__'s host part, restated) ----
st = sp.states
evs = np.flatnonzero(st["ev"] >= 0)
edist_first = np.full(len(st), -1, np.int64); n_ed = 0; cands = []
for ns in evs:
    r = st[ns]; ev_, cf = int(r["ev"]), int(r["call_first"]); k = int(r["call_pass0"]) + int(r["call_pass1"])
    edist_first[ns] = n_ed
    blk = np.zeros(3 + k, api.EVAL_CAND_DTYPE); evr = sp.ev_want[ev_]
    for c in range(3):
        d_ = sp.ev_dz[ev_, c]
        blk[c] = (d_["x"], d_["y"], d_["w"], d_["h"], d_["metric"], d_["qp"], c, 0,
                  int(evr["x"]) >> (1 if c else 0), int(evr["y"]) >> (1 if c else 0), 1, 0, sp.ev_weight[ev_, c])
    cc = sp.call_cand[cf:cf + k]; comp = sp.call_tx["comp"][cf:cf + k]; bb = blk[3:]
    for f in ("x", "y", "w", "h", "metric", "qp"): bb[f] = cc[f]
    bb["comp"], bb["versus"] = comp, 1
    bb["ox"], bb["oy"], bb["orig_at"] = int(evr["x"]) >> (comp != 0), int(evr["y"]) >> (comp != 0), 1
    bb["weight"] = sp.ev_weight[ev_][comp]; blk[3:] = bb; cands.append(blk); n_ed += 3 + k
allc = np.concatenate(cands) if cands else np.zeros(0, api.EVAL_CAND_DTYPE)
allc2 = allc.copy(); allc2["orig_at"] = 0
ce = sp.ev_want[sp.call_ev] if len(sp.call_tx) else sp.ev_want[:0]
sh = (sp.call_tx["comp"] != 0).astype(np.int64)
pos = np.zeros((len(sp.call_tx), 2), api.POS_DTYPE); pos["x"][:, 0], pos["y"][:, 0] = ce["x"] >> sh, ce["y"] >> sh
mc = np.zeros((len(sp.mg_inter), 5), api.EVAL_CAND_DTYPE)
for f in ("x", "y", "w", "h", "metric"): mc[f] = sp.mg_cands[f]
mc["ox"], mc["oy"], mc["orig_at"], mc["weight"] = sp.mg_want["x"][:, None], sp.mg_want["y"][:, None], 1, 1.0
ac = np.zeros(len(sp.aff_start_cands), api.EVAL_CAND_DTYPE)
for f in ("x", "y", "w", "h", "metric"): ac[f] = sp.aff_start_cands[f]
if len(ac): ac["ox"], ac["oy"] = sp.aff_start_copy["sx"], sp.aff_start_copy["sy"]
ac["orig_at"], ac["weight"] = 1, 1.0
# rebuild with ev_merge_slot known
inp = builder_inputs(sp, rl, True); b.destroy(); b = csb.Builder(inp)
for k, want in dict(ev_cands=allc, ev_cands_copy=allc2, edist_first=edist_first, call_pos=pos.reshape(-1),
                    mg_ecands=mc.reshape(-1), aff_start_ecands=ac).items():
    got = getattr(b, k)
    if got.tobytes() != np.ascontiguousarray(want).tobytes():
        bad += 1; print("MISMATCH", k, got.shape, want.shape)
print("n_edist", b.n_edist, n_ed, "bad", bad)

# ---- programs ----
class NS: pass
run = object.__new__(rs.ChainedRun)
run.sp, run.api = sp, api
base = [0x10000000]
def addr():
    base[0] += 0x10000000; return base[0]
t = NS()
for f in ("d_me d_me_res h_me_res d_bi d_bi_res h_bi_res d_bi_lic d_nb_copy d_mg_copy d_mg_inter d_mg_dst d_mg_cands "
          "d_ev_dst d_copy_orig d_call_copy_pred d_call_tx d_call_off d_call_prm d_contexts d_levels h_levels "
          "d_in_satd_jobs d_in_satd h_in_satd d_in_pred d_in_tx d_in_off d_in_nnz h_in_nnz d_in_contexts d_in_prm "
          "d_in_cand d_in_dist h_in_dist d_in_levels h_in_levels").split():
    setattr(t, f, addr())
run.t = t
run.d = {k: addr() for k in ("passes start_cands start_slots start_dist aff_start_inter aff_start_dst aff_start_cands "
                             "aff_start_copy aff_start_ecands me_work me_res_c me_slots aff_work aff_res_c aff_slots bi_work "
                             "bi_res_c bi_slots bi_lic_work ev_inter_work results mg_fold mg_slots mg_ecands ev_cands "
                             "ev_cands_copy call_pos").split()}
run.cres = {"results": (None, addr()), "ev_inter_out": (None, addr())}
run.z = {k: (None, addr()) for k in ("nnz", "edist", "mg_dist", "mg_res", "mg_slots_out")}
run.edist_first = edist_first
A = csb.Addrs()
for f in csb.ADDR_FIELDS:
    if hasattr(t, f): v = getattr(t, f)
    elif f in run.d: v = run.d[f]
    elif f == "h_results": v = run.cres["results"][1]
    elif f == "h_ev_inter_out": v = run.cres["ev_inter_out"][1]
    elif f.startswith("z_"): v = run.z[f[2:]][1]
    else: raise KeyError(f)
    setattr(A, f, v)
keep = dict(in_stage=np.ascontiguousarray(sp.in_stage, np.int32), in_ctx=np.ascontiguousarray(sp.in_ctx, np.int32),
            in_comp=np.ascontiguousarray(sp.in_comp, np.int32), in_weight=np.ascontiguousarray(sp.in_weight, np.float64),
            in_off=np.ascontiguousarray(sp.in_off, np.uint32), bi_ref=np.ascontiguousarray(sp.bi_ref, np.int8))
IN = csb.Intra()
for k, v in keep.items(): setattr(IN, k, v.ctypes.data if v.size else None)
IN.n_in, IN.n_in_levels = len(sp.in_off), int(sp.n_in_levels)
N = len(st)
pbad = 0
for (byp, ver, refs, live, nc, fe, mf) in [(1,1,1,0,1,1,1), (0,1,1,0,1,1,1), (1,0,1,1,1,1,1), (1,1,0,0,0,0,0), (1,1,1,0,1,0,1),
                                            (1,1,0,1,0,0,1), (1,0,1,0,0,0,0), (0,0,1,1,1,0,0)]:
    run.no_copies, run.fused_eval, run.merge_fold, run.refs_form = bool(nc), bool(fe), bool(mf), bool(refs)
    for first, n in ((0, N), (N // 3, min(2000, N - N // 3))):
        want = run.program(first, n, by_position=bool(byp), verify=bool(ver), refs_form=bool(refs), live=bool(live))
        flags = byp * csb.BY_POSITION | ver * csb.VERIFY | refs * csb.REFS_FORM | live * csb.LIVE | nc * csb.NO_COPIES | fe * csb.FUSED_EVAL | mf * csb.MERGE_FOLD
        got = b.program(A, IN, first, n, flags)
        ok = got.tobytes() == np.ascontiguousarray(want).tobytes()
        if not ok:
            pbad += 1
            print("PROGRAM MISMATCH", (byp, ver, refs, live, nc, fe, mf), first, n, len(got), len(want))
            m = min(len(got), len(want))
            d = [i for i in range(m) if got[i].tobytes() != want[i].tobytes()][:3]
            for i in d: print("   op", i, got[i], want[i])
print("programs bad:", pbad)
