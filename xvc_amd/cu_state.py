"""The CU-state walk from Python: the tables, environment and result arrays of one chain on
the device, and the calls into the C++ host layer that walks them (libxvchost.so:
xvc_amd/host/xvc_cu_state.cc - the serial form, one device chain per state, several chains
interleaved, the engine) with the op programs of xvc_gpu::CuStateBuilder
(xvc_amd/host/xvc_cu_state_builder.cc through xvc_amd/cu_state_builder.py).

A walk is built from a PICTURE TABLE object: the job arrays of the C-ABI entry points for
all states of a picture in issue order (me_jobs, bi_jobs, aff_jobs, mg_*, ev_*, call_*,
in_*, nb_* ...), the xvc_cs_state records (STATE_DTYPE) and, for the chained form, the
builder's outputs.  Where the arrays come from is the caller's business: an encoder fills
them as CuEncoder::CompressCu (cu_encoder.cc:123-273) reaches each state; the test harness
(tests/rd_serial.py) fills them from a captured encode and compares what comes back.

States (the `kind` of a record):
  0  merge ranking   SearchMergeCandidates (inter_search.cc:165-197)
  1  evaluation      CompressAndEvalCbf (:261-365) of a given motion
  2  inter mode      CompressInter (:74-98): SearchMotion [+ affine] then its evaluation
  3  motion only     a CompressInter that returned before its evaluation (:94-96)
  4  intra mode      CompressIntra (cu_encoder.cc:518-541)"""
import ctypes as C

import numpy as np

from . import cu_state_builder as csb

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


class Walk:
    """One chain: a context (its own stream), the picture's job arrays on the device,
    scratch pictures and result arrays of its own.  sp: the picture table (module
    docstring); pics: POC -> device picture of the reference pictures; orig_planes: the
    picture being encoded, padded by `border` samples (Picture.upload's arguments)."""

    def __init__(self, api, ctx, sp, pics, width, height, orig_planes, border):
        from . import decoder
        self.api, self.ctx, self.sp = api, ctx, sp
        self.lib = decoder.load_host_library()
        for f in ("xvc_host_cu_state_run_serial",):
            getattr(self.lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p]
        self.orig = ctx.picture(width, height, 10)
        self.orig.upload(orig_planes, border)
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

    def destroy(self):
        for p in self._pinned:
            self.ctx.lib.xvcgpu_host_free(self.ctx.h, p)
        for b in self._keep:
            if hasattr(b, "free"):
                b.free()
        for p in self.scratch + [self.orig, self.rec, self.nb, self.ipred, self.irec]:
            p.destroy()



class CsEnv(C.Structure):
    _fields_ = [("orig", C.c_void_p), ("refs", C.c_void_p), ("n_refs", C.c_int32),
                ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("reserved", C.c_int32),
                ("s_orig", C.c_void_p), ("s_pred", C.c_void_p), ("s_rec", C.c_void_p),
                ("d_levels", C.c_void_p), ("d_results", C.c_void_p),
                ("rec", C.c_void_p), ("nb", C.c_void_p), ("ipred", C.c_void_p), ("irec", C.c_void_p),
                ("d_in_levels", C.c_void_p)]


class ChainedWalk(Walk):
    """Walk + the arrays and the program of the chained form.  The picture table carries
    the outputs of xvc_gpu::CuStateBuilder (sp.builder and its arrays)."""

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

    def __init__(self, api, ctx, sp, pics, width, height, orig_planes, border):
        super().__init__(api, ctx, sp, pics, width, height, orig_planes, border)
        if not hasattr(sp, "builder"):
            raise ValueError("the picture table has not been through CuStateBuilder")
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
