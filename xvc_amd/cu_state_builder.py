"""ctypes binding of xvc_gpu::CuStateBuilder (xvc_amd/host/xvc_cu_state_builder.{h,cc}): the
composer of the CU-state walk.  The caller fills the input records (what the encoder's
control code holds at a CU state); the arrays and programs come back as numpy views."""
import ctypes as C

import numpy as np

from . import api

R3 = 3
ICTX_DTYPE = np.dtype([("merge_flag", "u1"), ("inter_dir_bi", "u1"), ("inter_dir_l", "u1"),
                       ("affine_flag", "u1"), ("ref_idx", "u1", (2,)), ("mvd", "u1", (2,)),
                       ("mvp_idx", "u1"), ("fullpel_mv", "u1"), ("lic_flag", "u1"), ("flags", "u1"),
                       ("num_refs", "u1", (2,)), ("frac_bits", "<u2")])
REF_ENTRY_DTYPE = np.dtype([("list", "i1"), ("ref_idx", "i1"), ("reused", "u1"), ("reserved", "u1"),
                            ("mvp", "<i4", (2, 3, 2))])
PASS_IN_DTYPE = np.dtype([("first", "<i4"), ("n", "<i4"), ("lambda16", "<u4"), ("fullpel", "u1"),
                          ("reserved", "u1", (3,)), ("ictx", ICTX_DTYPE)])
MOTION_DTYPE = np.dtype([("state", "<i4"), ("nb", "<i4"), ("plain", PASS_IN_DTYPE),
                         ("affine", PASS_IN_DTYPE)])
NEIGHBOURS_DTYPE = np.dtype([("has_above", "u1"), ("has_left", "u1"), ("above_x", "<i2"),
                             ("above_y", "<i2"), ("left_x", "<i2"), ("left_y", "<i2")])
MERGE_DTYPE = np.dtype([("lambda_sqrt", "<f8"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("any_lic", "u1"), ("reserved", "u1"), ("nb", "<i4"), ("state", "<i4")],
                       align=True)
EVAL_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("state", "<i4"), ("merge_slot", "<i4"),
                       ("dz", api.CAND_DTYPE, (3,)), ("weight", "<f8", (3,))], align=True)
assert (REF_ENTRY_DTYPE.itemsize, PASS_IN_DTYPE.itemsize, MOTION_DTYPE.itemsize,
        NEIGHBOURS_DTYPE.itemsize, MERGE_DTYPE.itemsize, EVAL_DTYPE.itemsize) == (52, 32, 72, 10, 24, 72)

PASS_DTYPE = np.dtype([
    ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("flags", "u1"), ("num_refs", "u1", 2),
    ("same_poc_in_l0", "i1", R3), ("lambda16", "<u4"), ("ictx", ICTX_DTYPE),
    ("mvp", "<i4", (2, R3, 2, 3, 2)), ("uni_job", "<i4", (2, R3)), ("start_dist", "<i4", (2, R3)),
    ("prev_job", "<i4", (2, R3)), ("bi_job", "<i4"), ("plain_pass", "<i4"), ("eval", "<i4"),
    ("slot", "i1", (2, R3)), ("bi_iterations", "u1"), ("reserved", "u1")], align=True)
MERGE_FOLD_DTYPE = np.dtype([("lambda_sqrt", "<f8"), ("dist", "<i4"), ("cand", "<i4"), ("slot", "<i4"),
                             ("reserved", "<i4")], align=True)
OP_DTYPE = np.dtype([("opcode", "<i4"), ("n", "<i4"), ("r0", "<i4"), ("r1", "<i4"), ("i0", "<i4"),
                     ("reserved", "<i4"), ("f", "<f8"), ("p", "<u8", 8)], align=True)

# XVC_CSB_* array ids and their element types
ARRAYS = [("passes", PASS_DTYPE), ("pass_first", np.dtype("<i8")), ("pass_count", np.dtype("<i8")),
          ("folded", np.dtype("u1")), ("start_cands", api.MCM_DTYPE), ("start_slots", np.dtype("u1")),
          ("aff_start_inter", api.INTER_DTYPE), ("aff_start_dst", api.POS_DTYPE),
          ("aff_start_cands", api.CAND_DTYPE), ("aff_start_copy", api.COPY_BLOCK_DTYPE),
          ("me_work", api.ME_DTYPE), ("bi_lic_work", api.LIC_DTYPE), ("aff_work", api.AFFINE_ME_DTYPE),
          ("aff_work_src", np.dtype("<i8")), ("me_slots", np.dtype("u1")), ("bi_slots", np.dtype("u1")),
          ("aff_slots", np.dtype("u1")), ("ev_inter_work", api.INTER_DTYPE),
          ("mg_fold", MERGE_FOLD_DTYPE), ("mg_slots", api.INTER_DTYPE), ("merge_state", np.dtype("<i8")),
          ("ev_cands", api.EVAL_CAND_DTYPE), ("ev_cands_copy", api.EVAL_CAND_DTYPE),
          ("edist_first", np.dtype("<i8")), ("call_pos", api.POS_DTYPE),
          ("mg_ecands", api.EVAL_CAND_DTYPE), ("aff_start_ecands", api.EVAL_CAND_DTYPE)]
(BY_POSITION, VERIFY, REFS_FORM, LIVE, NO_COPIES, FUSED_EVAL, MERGE_FOLD) = (1, 2, 4, 8, 16, 32, 64)

ADDR_FIELDS = (
    "d_me d_me_res h_me_res d_bi d_bi_res h_bi_res d_bi_lic "
    "d_nb_copy d_mg_copy d_mg_inter d_mg_dst d_mg_cands "
    "d_ev_dst d_copy_orig d_call_copy_pred d_call_tx d_call_off d_call_prm d_contexts "
    "d_levels h_levels "
    "d_in_satd_jobs d_in_satd h_in_satd d_in_pred d_in_tx d_in_off d_in_nnz h_in_nnz "
    "d_in_contexts d_in_prm d_in_cand d_in_dist h_in_dist d_in_levels h_in_levels "
    "passes start_cands start_slots start_dist aff_start_inter aff_start_dst "
    "aff_start_cands aff_start_copy aff_start_ecands me_work me_res_c me_slots "
    "aff_work aff_res_c aff_slots bi_work bi_res_c bi_slots bi_lic_work "
    "ev_inter_work results h_results h_ev_inter_out "
    "mg_fold mg_slots mg_ecands z_mg_dist z_mg_res z_mg_slots_out "
    "ev_cands ev_cands_copy call_pos z_nnz z_edist").split()


class Addrs(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ADDR_FIELDS]


class Intra(C.Structure):
    _fields_ = [("in_stage", C.c_void_p), ("in_ctx", C.c_void_p), ("in_comp", C.c_void_p),
                ("in_weight", C.c_void_p), ("in_off", C.c_void_p), ("n_in", C.c_int32),
                ("n_in_levels", C.c_int64), ("bi_ref", C.c_void_p)]


class Picture(C.Structure):
    _fields_ = [("states", C.c_void_p), ("n_states", C.c_int32),
                ("ref_poc", C.c_int32 * R3 * 2), ("n_ref", C.c_int32 * 2),
                ("slot_pocs", C.c_void_p), ("n_slots", C.c_int32), ("lic_folds", C.c_int32),
                ("motions", C.c_void_p), ("n_motions", C.c_int32),
                ("entries", C.c_void_p), ("nb", C.c_void_p),
                ("me_jobs", C.c_void_p), ("me_ref", C.c_void_p), ("n_me", C.c_int32),
                ("aff_jobs", C.c_void_p), ("aff_ref", C.c_void_p), ("n_aff", C.c_int32),
                ("ev_inter", C.c_void_p), ("n_ev", C.c_int32),
                ("merges", C.c_void_p), ("n_merges", C.c_int32),
                ("evals", C.c_void_p), ("ev_ctx", C.c_void_p),
                ("call_cand", C.c_void_p), ("call_comp", C.c_void_p), ("call_ev", C.c_void_p),
                ("n_calls", C.c_int32), ("mg_cands", C.c_void_p)]


def _lib():
    lib = api.load_host_library() if hasattr(api, "load_host_library") else None
    if lib is None:
        import os
        lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libxvchost.so"))
    lib.xvc_host_csb_build.argtypes = [C.c_void_p, C.c_void_p]
    lib.xvc_host_csb_destroy.argtypes = [C.c_void_p]
    lib.xvc_host_csb_array.restype = C.c_void_p
    lib.xvc_host_csb_array.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.xvc_host_csb_n_start_dist.argtypes = [C.c_void_p]
    lib.xvc_host_csb_n_bi_slots.argtypes = [C.c_void_p]
    lib.xvc_host_csb_n_edist.restype = C.c_int64
    lib.xvc_host_csb_n_edist.argtypes = [C.c_void_p]
    lib.xvc_host_csb_program.restype = C.c_void_p
    lib.xvc_host_csb_program.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p]
    return lib


class Builder:
    """Owns a xvc_csb and the input arrays it points into."""

    def __init__(self, inputs):
        """inputs: dict of numpy arrays / scalars named as the fields of xvc_csb_picture
        (ref_lists = ([pocs of list 0], [pocs of list 1]))."""
        self.lib = _lib()
        self._keep = {}
        pic = Picture()

        def arr(name, dt):
            a = np.ascontiguousarray(inputs[name], dtype=dt)
            self._keep[name] = a
            return a.ctypes.data if a.size else None

        st = inputs["states"]
        pic.states, pic.n_states = arr("states", st.dtype), len(st)
        for l in range(2):
            pic.n_ref[l] = len(inputs["ref_lists"][l])
            for r, poc in enumerate(inputs["ref_lists"][l]):
                pic.ref_poc[l][r] = int(poc)
        pic.slot_pocs, pic.n_slots = arr("slot_pocs", np.int32), len(inputs["slot_pocs"])
        pic.lic_folds = int(bool(inputs["lic_folds"]))
        pic.motions, pic.n_motions = arr("motions", MOTION_DTYPE), len(inputs["motions"])
        pic.entries = arr("entries", REF_ENTRY_DTYPE)
        pic.nb = arr("nb", NEIGHBOURS_DTYPE)
        pic.me_jobs, pic.me_ref, pic.n_me = arr("me_jobs", api.ME_DTYPE), arr("me_ref", np.int8), len(inputs["me_jobs"])
        pic.aff_jobs, pic.aff_ref = arr("aff_jobs", api.AFFINE_ME_DTYPE), arr("aff_ref", np.int8)
        pic.n_aff = len(inputs["aff_jobs"])
        pic.ev_inter, pic.n_ev = arr("ev_inter", api.INTER_DTYPE), len(inputs["evals"])
        pic.merges, pic.n_merges = arr("merges", MERGE_DTYPE), len(inputs["merges"])
        pic.evals, pic.ev_ctx = arr("evals", EVAL_DTYPE), arr("ev_ctx", np.int32)
        pic.call_cand, pic.call_comp = arr("call_cand", api.CAND_DTYPE), arr("call_comp", np.uint8)
        pic.call_ev, pic.n_calls = arr("call_ev", np.int32), len(inputs["call_cand"])
        pic.mg_cands = arr("mg_cands", api.CAND_DTYPE)
        self.h = C.c_void_p()
        rc = self.lib.xvc_host_csb_build(C.byref(pic), C.byref(self.h))
        if rc:
            raise ValueError("xvc_host_csb_build: %d (an input the device folds do not run)" % rc)
        self.n_start_dist = self.lib.xvc_host_csb_n_start_dist(self.h)
        self.n_bi_slots = self.lib.xvc_host_csb_n_bi_slots(self.h)
        self.n_edist = self.lib.xvc_host_csb_n_edist(self.h)
        for i, (name, dt) in enumerate(ARRAYS):
            nb = C.c_int64()
            p = self.lib.xvc_host_csb_array(self.h, i, C.byref(nb))
            if nb.value:
                a = np.frombuffer((C.c_char * nb.value).from_address(p), dt).copy()
            else:
                a = np.zeros(0, dt)
            setattr(self, name, a)

    def program(self, addrs, intra, first, n, flags):
        n_ops = C.c_int64()
        p = self.lib.xvc_host_csb_program(self.h, C.byref(addrs), C.byref(intra), first, n, flags,
                                          C.byref(n_ops))
        if not n_ops.value:
            return np.zeros(0, OP_DTYPE)
        return np.frombuffer((C.c_char * (n_ops.value * OP_DTYPE.itemsize)).from_address(p),
                             OP_DTYPE).copy()

    def destroy(self):
        if self.h:
            self.lib.xvc_host_csb_destroy(self.h)
            self.h = None
