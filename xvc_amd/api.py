"""Python host side of libxvcgpu.so (ctypes over the C-ABI in include/xvcgpu.h).

This is plumbing for tests and bench.py: it mirrors the C-ABI one-to-one and
adds numpy conveniences (host arrays in, host arrays out).  There is no CPU
fallback: if the HIP library is missing or no gfx950 device is present the
constructors raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# XVCGPU_LIB: developer knob to load an alternative build of the same library
# (kernel tuning experiments); always a HIP build, never a CPU path.
LIB_PATH = os.environ.get("XVCGPU_LIB") or os.path.join(HERE, "libxvcgpu.so")

BORDER_LUMA = 128
BORDER_CHROMA = 64
ME_FULLPEL = 1
ME_SUBPEL = 2
ME_LIC_JOBS = 4      # the batch holds XVC_ME_USE_LIC jobs (fullpel_mv bit 1)
ME_HINT_SQ16 = 8     # performance hint: (almost) all jobs are 16x16 / 16x8 CUs (xvcgpu.h)
ME_ONLY_SQ16 = 16    # the caller's word: every job is one (another shape: answered unsupported)

METRIC_SSD, METRIC_SATD, METRIC_SATD_ACONLY, METRIC_SAD, METRIC_SAD_FAST, \
    METRIC_SAD_ACONLY, METRIC_SAD_ACONLY_FAST, METRIC_STRUCTURAL_SSD = range(8)
TX_DEFAULT, TX_DCT2, TX_DCT5, TX_DCT8, TX_DST1, TX_DST7 = range(6)

CU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"),
                     ("intra", "u1"), ("cbf_luma", "u1"), ("qp_y", "i1"),
                     ("qp_c", "i1"), ("ref_idx0", "i1"), ("reserved", "i1"),
                     ("ref_poc", "<i4", (2,)), ("mv", "<i4", (2, 4, 2))])
ME_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("depth_nonzero", "u1"), ("fullpel_mv", "u1"),
                     ("mvp_x", "<i4"), ("mvp_y", "<i4"), ("prev_x", "<i4"),
                     ("prev_y", "<i4"), ("lambda16", "<u4"),
                     ("search_range", "<i4")])
MERES_DTYPE = np.dtype([("fullpel_x", "<i4"), ("fullpel_y", "<i4"),
                        ("mv_x", "<i4"), ("mv_y", "<i4"),
                        ("fullpel_cost", "<u4"), ("subpel_dist", "<u4")])
TX_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("comp", "u1"), ("tx_hor", "u1"), ("tx_ver", "u1"),
                     ("dst4x4", "u1"), ("qp", "i1"), ("intra_pic", "u1")])
MC_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                     ("comp", "u1"), ("reserved", "u1"), ("mv_x", "<i4"),
                     ("mv_y", "<i4")])
BI_DTYPE = np.dtype([("blk", ME_DTYPE), ("other_mv_x", "<i4"),
                     ("other_mv_y", "<i4"), ("boot_mv_x", "<i4"),
                     ("boot_mv_y", "<i4")])
MCBI_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                       ("comp", "u1"), ("reserved", "u1"), ("mv0_x", "<i4"),
                       ("mv0_y", "<i4"), ("mv1_x", "<i4"), ("mv1_y", "<i4")])
assert BI_DTYPE.itemsize == 48 and MCBI_DTYPE.itemsize == 24
MCAFF_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("comp", "u1"), ("reserved", "u1"), ("mv", "<i4", (3, 2))])
assert MCAFF_DTYPE.itemsize == 32
MCM_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                      ("metric", "u1"), ("qp", "i1"), ("mv_x", "<i4"), ("mv_y", "<i4")])
assert MCM_DTYPE.itemsize == 16
AFFINE_ME_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                            ("flags", "u1"), ("reserved", "u1"), ("lambda16", "<u4"),
                            ("mvp", "<i4", (3, 2)), ("bootstrap", "<i4", (3, 2)),
                            ("other_mv", "<i4", (3, 2))])
AFFINE_ME_RESULT_DTYPE = np.dtype([("mv", "<i4", (3, 2)), ("dist", "<u4"),
                                   ("iterations", "<u4")])
AFFINE_ME_HAS_BOOTSTRAP, AFFINE_ME_BIPRED = 1, 2
SEG_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("bytes", "<u8")])
LIC_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("comp", "u1"),
                      ("neighbors", "u1"), ("mv_x", "<i4"), ("mv_y", "<i4"),
                      ("above_x", "<i2"), ("above_y", "<i2"), ("left_x", "<i2"),
                      ("left_y", "<i2")])
assert LIC_DTYPE.itemsize == 24
INTRA_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("comp", "u1"), ("mode", "u1"), ("neighbors", "u1"),
                        ("above_right", "u1"), ("below_left", "u1"), ("reserved", "u1")])
assert INTRA_DTYPE.itemsize == 12
INTER_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("comp", "u1"),
                        ("flags", "u1"), ("ref", "i1", (2,)), ("neighbors", "u1"),
                        ("reserved", "u1"), ("above_x", "<i2"), ("above_y", "<i2"),
                        ("left_x", "<i2"), ("left_y", "<i2"), ("mv", "<i4", (2, 3, 2))])
assert INTER_DTYPE.itemsize == 68
INTER_AFFINE, INTER_LIC = 1, 2
POS_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2")])
COPY_BLOCK_DTYPE = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("dx", "<i2"), ("dy", "<i2"),
                             ("w", "u1"), ("h", "u1"), ("comp", "u1"), ("reserved", "u1")])
assert COPY_BLOCK_DTYPE.itemsize == 12
RDOQ_CTX_DTYPE = np.dtype([
    ("csbf", "u1", (2, 2)), ("sig_luma", "u1", (54,)), ("sig_chroma", "u1", (12,)),
    ("greater1_luma", "u1", (16,)), ("greater1_chroma", "u1", (6,)),
    ("last_x_luma", "u1", (25,)), ("last_y_luma", "u1", (25,)), ("last_x_chroma", "u1", (3,)),
    ("last_y_chroma", "u1", (3,)), ("cbf_luma", "u1"), ("cbf_chroma", "u1"), ("root_cbf", "u1"),
    ("reserved", "u1")])
RDOQ_PARAMS_DTYPE = np.dtype([("lambda", "<i8"), ("rd_factor", "<i8"), ("ctx_index", "<u2"),
                              ("flags", "u1"), ("reserved", "u1", (5,))])
assert RDOQ_CTX_DTYPE.itemsize == 152 and RDOQ_PARAMS_DTYPE.itemsize == 24
RDOQ_INTRA_CU, RDOQ_NO_2X2 = 1, 2
TXF_RDOQ = 16
INTRA_NUM_MODES = 67
INTRA_HAS_ABOVE_LEFT, INTRA_HAS_ABOVE, INTRA_HAS_LEFT = 1, 2, 4
# xvcgpu_tx_block.intra_pic flag bits (include/xvcgpu_types.h XVC_TXF_*)
TXF_INTRA_PIC, TXF_NO_SIGN_HIDING, TXF_SCAN_SHIFT = 1, 2, 2
TXE_ALT_DTYPE = np.dtype([("dist_reco", "<u8"), ("dist_resi", "<u8"), ("bits", "<u4"),
                          ("kind", "u1"), ("cbf", "u1"), ("reserved", "u1", (2,))])
TXE_JOB_DTYPE = np.dtype([("lambda", "<f8"), ("prev_cost", "<u8"), ("dist_zero", "<u8"),
                          ("bits_zero", "<u4"), ("alt_first", "<u4"), ("n_alt", "u1"),
                          ("flags", "u1"), ("reserved", "u1", (6,))])
TXE_RESULT_DTYPE = np.dtype([("cost", "<u8"), ("dist_reco", "<u8"), ("dist_resi", "<u8"),
                             ("best", "<i4"), ("cbf", "u1"), ("reserved", "u1", (3,))])
ROOT_CBF_JOB_DTYPE = np.dtype([("lambda", "<f8"), ("dist_resi", "<u8", (3,)),
                               ("dist_reco", "<u8", (3,)), ("dist_zero", "<u8", (3,)),
                               ("best_cu_cost", "<u8"), ("bits_non_zero", "<u4"),
                               ("bits_root_zero", "<u4"), ("bits_full", "<u4"),
                               ("cbf", "u1", (3,)), ("flags", "u1")])
ROOT_CBF_RESULT_DTYPE = np.dtype([("sum_dist_final", "<u8"), ("sum_dist_resi", "<u8"),
                                  ("root_cbf", "u1"), ("second_pass", "u1"),
                                  ("reserved", "u1", (6,))])
assert (TXE_ALT_DTYPE.itemsize, TXE_JOB_DTYPE.itemsize, TXE_RESULT_DTYPE.itemsize,
        ROOT_CBF_JOB_DTYPE.itemsize, ROOT_CBF_RESULT_DTYPE.itemsize) == (24, 40, 32, 104, 24)
TXE_KIND_NORMAL, TXE_KIND_TSKIP, TXE_KIND_SELECT = 0, 1, 2
TXE_CBF_ZERO, TXE_FAST_SELECT, TXE_PREV_CBF = 1, 2, 4
TXE_DIST_INVALID = 0xffffffffffffffff
EVAL_CAND_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"), ("metric", "u1"),
                            ("qp", "i1"), ("comp", "u1"), ("versus", "u1"), ("ox", "<i2"),
                            ("oy", "<i2"), ("orig_at", "u1"), ("reserved", "u1"), ("weight", "<f8")])
assert EVAL_CAND_DTYPE.itemsize == 24
CAND_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
                       ("metric", "u1"), ("qp", "i1"), ("mv_x", "<i2"),
                       ("mv_y", "<i2")])
assert CU_DTYPE.itemsize == 84 and ME_DTYPE.itemsize == 32
assert MERES_DTYPE.itemsize == 24 and TX_DTYPE.itemsize == 12
assert MC_DTYPE.itemsize == 16 and CAND_DTYPE.itemsize == 12

class FramePassArgs(C.Structure):
    """xvcgpu_frame_pass_args (include/xvcgpu_types.h)"""
    _fields_ = [("orig", C.c_void_p), ("ref", C.c_void_p), ("rec", C.c_void_p),
                ("d_me", C.c_void_p), ("d_results", C.c_void_p),
                ("n_cus", C.c_int32), ("max_block_size", C.c_int32),
                ("qp_y", C.c_int32), ("qp_c", C.c_int32), ("ref_poc", C.c_int32),
                ("d_nnz", C.c_void_p), ("d_cus_own", C.c_void_p), ("d_cus", C.c_void_p),
                ("n_cus_total", C.c_int32), ("d_cu_map", C.c_void_p),
                ("map_stride", C.c_int32), ("db_y_begin", C.c_int32),
                ("db_y_end", C.c_int32), ("dbh_y_end", C.c_int32),
                ("ssd_y_begin", C.c_int32), ("ssd_y_end", C.c_int32),
                ("shift_bitdepth", C.c_int32), ("d_ssd", C.c_void_p),
                ("d_rdoq_contexts", C.c_void_p), ("d_rdoq_params", C.c_void_p),
                ("pred", C.c_void_p), ("d_tx", C.c_void_p), ("d_level_off", C.c_void_p),
                ("d_luma_tx_index", C.c_void_p), ("d_coeffs", C.c_void_p),
                ("d_levels", C.c_void_p), ("n_tx", C.c_int32), ("n_coeffs", C.c_uint32),
                ("scratch_rec", C.c_void_p), ("tx_four_lane_only", C.c_int32), ("me_only_sq16", C.c_int32)]


FP_ENCODE, FP_DEBLOCK_V, FP_DEBLOCK_H, FP_PAD, FP_SSD = 1, 2, 4, 8, 16

# every symbol include/xvcgpu.h declares
SYMBOLS = [
    "xvcgpu_create", "xvcgpu_destroy", "xvcgpu_last_error", "xvcgpu_version",
    "xvcgpu_set_stream", "xvcgpu_get_stream", "xvcgpu_use_own_stream", "xvcgpu_use_priority_stream", "xvcgpu_set_short_kernel_priority", "xvcgpu_wait_for", "xvcgpu_sync", "xvcgpu_timer_begin", "xvcgpu_timer_end", "xvcgpu_timer_mark", "xvcgpu_timer_between",
    "xvcgpu_record_begin", "xvcgpu_record_end", "xvcgpu_replay", "xvcgpu_recording_destroy",
    "xvcgpu_malloc", "xvcgpu_free", "xvcgpu_memcpy_h2d", "xvcgpu_memcpy_d2h", "xvcgpu_memcpy_d2h_async", "xvcgpu_upload_ahead", "xvcgpu_eval_dist_batch", "xvcgpu_cs_start_fold", "xvcgpu_cs_uni_fold", "xvcgpu_cs_bi_fold", "xvcgpu_cs_merge_fold", "xvcgpu_residual_rdoq_batch_at", "xvcgpu_quant_rdo_set_four_lane_only", "xvcgpu_cs_env_create", "xvcgpu_cs_env_destroy", "xvcgpu_cs_segs_launch", "xvcgpu_event_query",
    "xvcgpu_memset", "xvcgpu_picture_create", "xvcgpu_picture_bytes",
    "xvcgpu_picture_wrap", "xvcgpu_picture_destroy", "xvcgpu_picture_upload",
    "xvcgpu_picture_download", "xvcgpu_picture_upload_padded",
    "xvcgpu_picture_download_padded", "xvcgpu_picture_plane",
    "xvcgpu_picture_copy", "xvcgpu_pad_border", "xvcgpu_metric_batch", "xvcgpu_mc_metric_batch",
    "xvcgpu_me_search", "xvcgpu_me_search_sized", "xvcgpu_mc_batch", "xvcgpu_mc_from_me",
    "xvcgpu_mc_bipred_batch", "xvcgpu_bipred_search", "xvcgpu_mc_affine_batch", "xvcgpu_mc_lic_batch",
    "xvcgpu_affine_me_batch",
    "xvcgpu_me_search_refs",
    "xvcgpu_bipred_search_refs",
    "xvcgpu_mc_metric_batch_refs",
    "xvcgpu_affine_me_batch_refs",
    "xvcgpu_cu_info_from_me", "xvcgpu_recon_from_me", "xvcgpu_residual_batch",
    "xvcgpu_fwd_transform_batch", "xvcgpu_inv_transform_batch",
    "xvcgpu_deblock", "xvcgpu_deblock_rows", "xvcgpu_deblock_pad_ssd", "xvcgpu_picture_ssd", "xvcgpu_picture_ssd_rows",
    "xvcgpu_picture_import", "xvcgpu_picture_export", "xvcgpu_picture_crc",
    "xvcgpu_variance_map", "xvcgpu_histogram_distance",
    "xvcgpu_intra_pred_batch", "xvcgpu_intra_satd_batch", "xvcgpu_intra_recon_batch",
    "xvcgpu_intra_select_modes", "xvcgpu_frame_pass", "xvcgpu_frame_pass_multi", "xvcgpu_copy_segments",
    "xvcgpu_get_transform_matrix", "xvcgpu_inter_pred_batch", "xvcgpu_deblock_tree",
    "xvcgpu_residual_rdoq_batch", "xvcgpu_quant_rdo_batch", "xvcgpu_recon_from_me_rdoq",
    "xvcgpu_quant_rdo_reserve", "xvcgpu_quant_rdo_class_counts", "xvcgpu_quant_rdo_set_prove_zero",
    "xvcgpu_tx_eval_batch", "xvcgpu_root_cbf_batch", "xvcgpu_bipred_search_lic",
    "xvcgpu_inter_pred_batch_to", "xvcgpu_copy_blocks", "xvcgpu_intra_recon_waves",
    "xvcgpu_host_alloc", "xvcgpu_host_free", "xvcgpu_memcpy_h2d_async",
    "xvcgpu_inv_transform_cu_order",
    "xvcgpu_fwd_from_me_classify", "xvcgpu_fwd_from_me_classify_prove", "xvcgpu_quant_rdo_classified_batch",
    "xvcgpu_event_create", "xvcgpu_event_destroy", "xvcgpu_event_record", "xvcgpu_event_wait",
    "xvcgpu_event_synchronize", "xvcgpu_comm_unique_id", "xvcgpu_comm_create",
    "xvcgpu_comm_destroy", "xvcgpu_comm_world", "xvcgpu_comm_rank", "xvcgpu_comm_wait_event",
    "xvcgpu_comm_record_event", "xvcgpu_comm_sync", "xvcgpu_comm_group_begin",
    "xvcgpu_comm_group_end", "xvcgpu_comm_send_picture", "xvcgpu_comm_recv_picture",
    "xvcgpu_comm_send_rows", "xvcgpu_comm_recv_rows", "xvcgpu_comm_all_reduce_sum_u64",
    "xvcgpu_comm_send_bytes", "xvcgpu_comm_recv_bytes", "xvcgpu_inv_transform_dist_batch",
    "xvcgpu_fwd_from_me",
]

_vp = C.c_void_p
_pd = C.c_ssize_t
_lib = None


class XvcGpuError(RuntimeError):
    pass


def load_library():
    """Load libxvcgpu.so (fails loudly; never substitutes a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same
    # SONAME as /opt/rocm's).  Importing torch first makes the dynamic linker
    # resolve libxvcgpu.so's dependency to that already-loaded copy, so torch
    # tensors, streams and RCCL share the runtime our kernels launch on.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise XvcGpuError(
            "libxvcgpu.so is not built: run `python -m xvc_amd.build` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.xvcgpu_version.restype = C.c_char_p
    lib.xvcgpu_last_error.restype = C.c_char_p
    lib.xvcgpu_last_error.argtypes = [_vp]
    lib.xvcgpu_picture_bytes.restype = C.c_size_t
    lib.xvcgpu_picture_bytes.argtypes = [C.c_int, C.c_int]
    lib.xvcgpu_destroy.restype = None
    lib.xvcgpu_destroy.argtypes = [_vp]
    lib.xvcgpu_picture_destroy.restype = None
    lib.xvcgpu_picture_destroy.argtypes = [_vp]
    lib.xvcgpu_recording_destroy.argtypes = [_vp]
    lib.xvcgpu_recording_destroy.restype = None
    sigs = {
        "xvcgpu_create": [C.c_int, C.POINTER(_vp)],
        "xvcgpu_set_stream": [_vp, _vp],
        "xvcgpu_use_own_stream": [_vp],
        "xvcgpu_use_priority_stream": [_vp, C.c_int],
        "xvcgpu_set_short_kernel_priority": [_vp, C.c_int],
        "xvcgpu_wait_for": [_vp, _vp],
        "xvcgpu_sync": [_vp],
        "xvcgpu_timer_begin": [_vp],
        "xvcgpu_timer_end": [_vp, C.POINTER(C.c_float)],
        "xvcgpu_timer_mark": [_vp, C.c_int],
        "xvcgpu_timer_between": [_vp, C.c_int, C.c_int, C.POINTER(C.c_float)],
        "xvcgpu_record_begin": [_vp],
        "xvcgpu_record_end": [_vp, C.POINTER(_vp)],
        "xvcgpu_replay": [_vp, _vp],
        "xvcgpu_malloc": [_vp, C.c_size_t, C.POINTER(_vp)],
        "xvcgpu_free": [_vp, _vp],
        "xvcgpu_memcpy_h2d": [_vp, _vp, _vp, C.c_size_t],
        "xvcgpu_memcpy_d2h": [_vp, _vp, _vp, C.c_size_t],
        "xvcgpu_memset": [_vp, _vp, C.c_int, C.c_size_t],
        "xvcgpu_picture_create": [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
        "xvcgpu_picture_wrap": [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t,
                                C.POINTER(_vp)],
        "xvcgpu_picture_upload": [_vp, C.POINTER(_vp), C.POINTER(_pd)],
        "xvcgpu_picture_download": [_vp, C.POINTER(_vp), C.POINTER(_pd)],
        "xvcgpu_picture_upload_padded": [_vp, C.POINTER(_vp), C.POINTER(_pd), C.c_int],
        "xvcgpu_picture_download_padded": [_vp, C.POINTER(_vp), C.POINTER(_pd), C.c_int],
        "xvcgpu_picture_plane": [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_pd)],
        "xvcgpu_picture_copy": [_vp, _vp, _vp],
        "xvcgpu_pad_border": [_vp, _vp],
        "xvcgpu_metric_batch": [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, _vp,
                                C.c_int, _vp],
        "xvcgpu_mc_metric_batch": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp],
        "xvcgpu_me_search": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp],
        "xvcgpu_me_search_sized": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_mc_batch": [_vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_mc_from_me": [_vp, _vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_mc_bipred_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_mc_affine_batch": [_vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_bipred_search": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_cu_info_from_me": [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                   C.c_int, C.c_int, _vp],
        "xvcgpu_recon_from_me": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, _vp, _vp],
        "xvcgpu_residual_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
        "xvcgpu_fwd_transform_batch": [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp],
        "xvcgpu_inv_transform_batch": [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
        "xvcgpu_deblock": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                           C.c_int, C.c_int],
        "xvcgpu_deblock_rows": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
        "xvcgpu_picture_ssd": [_vp, _vp, _vp, C.c_int, C.c_int, _vp],
        "xvcgpu_picture_ssd_rows": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
        "xvcgpu_picture_import": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int],
        "xvcgpu_picture_export": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "xvcgpu_picture_crc": [_vp, _vp, C.c_int, _vp],
        "xvcgpu_variance_map": [_vp, _vp, _vp, C.c_int, _vp],
        "xvcgpu_histogram_distance": [_vp, _vp, _vp, _vp],
        "xvcgpu_mc_lic_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_affine_me_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp],
        "xvcgpu_me_search_refs": [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_bipred_search_refs": [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_mc_metric_batch_refs": [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp],
        "xvcgpu_affine_me_batch_refs": [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_intra_pred_batch": [_vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_intra_satd_batch": [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_intra_recon_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
        "xvcgpu_intra_select_modes": [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int],
        "xvcgpu_frame_pass": [_vp, C.POINTER(FramePassArgs), C.c_int],
        "xvcgpu_frame_pass_multi": [C.POINTER(_vp), C.POINTER(C.POINTER(FramePassArgs)), C.c_int,
                                    C.c_int],
        "xvcgpu_copy_segments": [_vp, _vp, C.c_int],
        "xvcgpu_get_transform_matrix": [C.c_int, C.c_int, _vp],
        "xvcgpu_inter_pred_batch": [_vp, C.POINTER(_vp), C.c_int, _vp, _vp, _vp, C.c_int],
        "xvcgpu_residual_rdoq_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp],
        "xvcgpu_recon_from_me_rdoq": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _vp, _vp, _vp, _vp],
        "xvcgpu_quant_rdo_reserve": [_vp, C.c_int, C.c_size_t],
        "xvcgpu_quant_rdo_class_counts": [_vp, _vp],
        "xvcgpu_quant_rdo_set_prove_zero": [_vp, C.c_int],
        "xvcgpu_quant_rdo_set_four_lane_only": [_vp, C.c_int],
        "xvcgpu_tx_eval_batch": [_vp, _vp, C.c_int, _vp, _vp],
        "xvcgpu_inter_pred_batch_to": [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_copy_blocks": [_vp, _vp, _vp, _vp, C.c_int],
        "xvcgpu_host_alloc": [_vp, C.c_size_t, C.POINTER(_vp)],
        "xvcgpu_host_free": [_vp, _vp],
        "xvcgpu_inv_transform_cu_order": [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
        "xvcgpu_memcpy_h2d_async": [_vp, _vp, _vp, C.c_size_t],
        "xvcgpu_memcpy_d2h_async": [_vp, _vp, _vp, C.c_size_t],
        "xvcgpu_upload_ahead": [_vp, _vp, _vp, C.c_size_t, _vp, _vp],
        "xvcgpu_eval_dist_batch": [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp],
        "xvcgpu_cs_start_fold": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int],
        "xvcgpu_cs_uni_fold": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp],
        "xvcgpu_cs_bi_fold": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp],
        "xvcgpu_cs_merge_fold": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp],
        "xvcgpu_residual_rdoq_batch_at": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                          C.c_int, _vp, C.c_int, _vp],
        "xvcgpu_intra_recon_waves": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
        "xvcgpu_bipred_search_lic": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int],
        "xvcgpu_root_cbf_batch": [_vp, _vp, C.c_int, _vp],
        "xvcgpu_quant_rdo_batch": [_vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_size_t, _vp, _vp,
                                   _vp, _vp],
        "xvcgpu_deblock_pad_ssd": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, _vp],
        "xvcgpu_deblock_tree": [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int],
        "xvcgpu_event_create": [_vp, C.POINTER(_vp)],
        "xvcgpu_event_record": [_vp, _vp],
        "xvcgpu_event_wait": [_vp, _vp],
        "xvcgpu_event_synchronize": [_vp],
        "xvcgpu_comm_unique_id": [_vp],
        "xvcgpu_comm_create": [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)],
        "xvcgpu_comm_world": [_vp],
        "xvcgpu_comm_rank": [_vp],
        "xvcgpu_comm_wait_event": [_vp, _vp],
        "xvcgpu_comm_record_event": [_vp, _vp],
        "xvcgpu_comm_sync": [_vp],
        "xvcgpu_comm_group_begin": [_vp],
        "xvcgpu_comm_group_end": [_vp],
        "xvcgpu_comm_send_picture": [_vp, _vp, C.c_int],
        "xvcgpu_comm_recv_picture": [_vp, _vp, C.c_int],
        "xvcgpu_comm_send_rows": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "xvcgpu_comm_recv_rows": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "xvcgpu_comm_all_reduce_sum_u64": [_vp, _vp, C.c_int],
        "xvcgpu_comm_send_bytes": [_vp, _vp, C.c_size_t, C.c_int],
        "xvcgpu_inv_transform_dist_batch": [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp],
        "xvcgpu_fwd_from_me": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp],
        "xvcgpu_fwd_from_me_classify": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp],
        "xvcgpu_fwd_from_me_classify_prove": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                              C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, _vp,
                                              _vp, _vp, _vp],
        "xvcgpu_quant_rdo_classified_batch": [_vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_size_t,
                                              _vp, _vp, _vp, _vp, _vp],
        "xvcgpu_comm_recv_bytes": [_vp, _vp, C.c_size_t, C.c_int],
    }
    lib.xvcgpu_event_destroy.restype = None
    lib.xvcgpu_event_destroy.argtypes = [_vp]
    lib.xvcgpu_comm_destroy.restype = None
    lib.xvcgpu_comm_destroy.argtypes = [_vp]
    for name, args in sigs.items():
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = args
    _lib = lib
    return lib


def transform_matrix(tx_type, size):
    """Host-side table query (works without a GPU)."""
    lib = load_library()
    out = np.zeros((size, size), np.int16)
    st = lib.xvcgpu_get_transform_matrix(tx_type, size, out.ctypes.data)
    if st != 0:
        return None
    return out


class DeviceBuffer:
    """Raw device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        ctx._check(ctx.lib.xvcgpu_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, max(arr.nbytes, 1))
        if arr.nbytes:
            ctx._check(ctx.lib.xvcgpu_memcpy_h2d(ctx.h, buf.ptr, arr.ctypes.data,
                                                 arr.nbytes))
        return buf

    def to_array(self, dtype, count):
        out = np.zeros(count, dtype)
        if out.nbytes:
            self.ctx._check(self.ctx.lib.xvcgpu_memcpy_d2h(
                self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.xvcgpu_free(self.ctx.h, self.ptr)
            self.ptr = None


class Event:
    """An ordering point between streams (xvcgpu_event_*)."""

    def __init__(self, ctx):
        self.ctx = ctx
        p = _vp()
        ctx._check(ctx.lib.xvcgpu_event_create(ctx.h, C.byref(p)))
        self.h = p.value

    def record(self, ctx=None):
        ctx = ctx or self.ctx
        ctx._check(ctx.lib.xvcgpu_event_record(ctx.h, self.h))

    def wait(self, ctx=None):
        """Work queued on `ctx` from now on starts after the event."""
        ctx = ctx or self.ctx
        ctx._check(ctx.lib.xvcgpu_event_wait(ctx.h, self.h))

    def synchronize(self):
        self.ctx._check(self.ctx.lib.xvcgpu_event_synchronize(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.xvcgpu_event_destroy(self.h)
            self.h = None


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes for xvcgpu_comm_create, made on one rank (ncclGetUniqueId)."""
    lib = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    st = lib.xvcgpu_comm_unique_id(buf)
    if st != 0:
        raise XvcGpuError("xvcgpu_comm_unique_id failed with status %d" % st)
    return bytes(buf)


class Comm:
    """The RCCL communicator of this process (one per GPU) with its own stream:
    point-to-point transfers of pictures and plane rows (xvcgpu_comm_*)."""

    def __init__(self, ctx, unique_id, world, rank):
        self.ctx, self.world, self.rank = ctx, world, rank
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        p = _vp()
        ctx._check(ctx.lib.xvcgpu_comm_create(ctx.h, buf, world, rank, C.byref(p)))
        self.h = p.value

    def _c(self, st):
        self.ctx._check(st)

    def wait_event(self, ev):
        self._c(self.ctx.lib.xvcgpu_comm_wait_event(self.h, ev.h))

    def record_event(self, ev):
        self._c(self.ctx.lib.xvcgpu_comm_record_event(self.h, ev.h))

    def sync(self):
        self._c(self.ctx.lib.xvcgpu_comm_sync(self.h))

    def group_begin(self):
        self._c(self.ctx.lib.xvcgpu_comm_group_begin(self.h))

    def group_end(self):
        self._c(self.ctx.lib.xvcgpu_comm_group_end(self.h))

    def send_picture(self, pic, dst):
        self._c(self.ctx.lib.xvcgpu_comm_send_picture(self.h, pic.h_pic, dst))

    def recv_picture(self, pic, src):
        self._c(self.ctx.lib.xvcgpu_comm_recv_picture(self.h, pic.h_pic, src))

    def send_rows(self, pic, y0, y1, dst, comp_mask=7):
        self._c(self.ctx.lib.xvcgpu_comm_send_rows(self.h, pic.h_pic, comp_mask, y0, y1, dst))

    def recv_rows(self, pic, y0, y1, src, comp_mask=7):
        self._c(self.ctx.lib.xvcgpu_comm_recv_rows(self.h, pic.h_pic, comp_mask, y0, y1, src))

    def send_bytes(self, dev_ptr, nbytes, dst):
        self._c(self.ctx.lib.xvcgpu_comm_send_bytes(self.h, dev_ptr, nbytes, dst))

    def recv_bytes(self, dev_ptr, nbytes, src):
        self._c(self.ctx.lib.xvcgpu_comm_recv_bytes(self.h, dev_ptr, nbytes, src))

    def all_reduce_sum_u64(self, dev_ptr, n):
        self._c(self.ctx.lib.xvcgpu_comm_all_reduce_sum_u64(self.h, dev_ptr, n))

    def destroy(self):
        if self.h:
            self.ctx.lib.xvcgpu_comm_destroy(self.h)
            self.h = None


class Picture:
    """Device twin of YuvPicture (padded 4:2:0 16-bit planes in HBM)."""

    def __init__(self, ctx, width, height, bitdepth=10, wrap_ptr=None,
                 wrap_bytes=0):
        self.ctx = ctx
        self.w, self.h, self.bd = width, height, bitdepth
        p = _vp()
        if wrap_ptr is None:
            ctx._check(ctx.lib.xvcgpu_picture_create(ctx.h, width, height, bitdepth,
                                                     C.byref(p)))
        else:
            ctx._check(ctx.lib.xvcgpu_picture_wrap(ctx.h, width, height, bitdepth,
                                                   wrap_ptr, wrap_bytes, C.byref(p)))
        self.h_pic = p.value

    def _args(self, planes, border=0):
        pp = (_vp * 3)()
        ss = (_pd * 3)()
        keep = []
        for c in range(3):
            a = planes[c]
            if a is None:
                pp[c] = None
                ss[c] = 0
                continue
            assert a.dtype == np.uint16 and a.strides[1] == 2
            b = border if c == 0 else border // 2
            pp[c] = a.ctypes.data + (b * a.strides[0] + b * 2)
            ss[c] = a.strides[0] // 2
            keep.append(a)
        return pp, ss, keep

    def nbytes(self):
        """Device bytes of the padded picture (what one transfer moves)."""
        return int(self.ctx.lib.xvcgpu_picture_bytes(self.w, self.h))

    def upload(self, planes, border=0):
        """planes: [Y,U,V] uint16 2-D arrays; with border>0 the arrays include
        `border` (luma) / border//2 (chroma) samples on every side."""
        pp, ss, keep = self._args(planes, border)
        lib = self.ctx.lib
        if border:
            self.ctx._check(lib.xvcgpu_picture_upload_padded(self.h_pic, pp, ss, border))
        else:
            self.ctx._check(lib.xvcgpu_picture_upload(self.h_pic, pp, ss))
        del keep

    def download(self, border=0):
        planes = []
        for c in range(3):
            b = border if c == 0 else border // 2
            w, h = (self.w, self.h) if c == 0 else (self.w // 2, self.h // 2)
            planes.append(np.zeros((h + 2 * b, w + 2 * b), np.uint16))
        pp, ss, keep = self._args(planes, border)
        lib = self.ctx.lib
        if border:
            self.ctx._check(lib.xvcgpu_picture_download_padded(self.h_pic, pp, ss, border))
        else:
            self.ctx._check(lib.xvcgpu_picture_download(self.h_pic, pp, ss))
        del keep
        return planes

    def plane_ptr(self, comp):
        p = _vp()
        s = _pd()
        self.ctx._check(self.ctx.lib.xvcgpu_picture_plane(self.h_pic, comp,
                                                          C.byref(p), C.byref(s)))
        return p.value, s.value

    def destroy(self):
        if self.h_pic:
            self.ctx.lib.xvcgpu_picture_destroy(self.h_pic)
            self.h_pic = None


class Context:
    def __init__(self, device=0):
        self.lib = load_library()
        self.device = device
        h = _vp()
        st = self.lib.xvcgpu_create(device, C.byref(h))
        if st != 0:
            raise XvcGpuError("xvcgpu_create failed with status %d "
                              "(no gfx950 device? there is no CPU fallback)" % st)
        self.h = h.value

    def _check(self, st):
        if st != 0:
            raise XvcGpuError("xvcgpu status %d: %s" % (
                st, self.lib.xvcgpu_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.lib.xvcgpu_destroy(self.h)
            self.h = None

    def sync(self):
        self._check(self.lib.xvcgpu_sync(self.h))

    def set_stream(self, hip_stream):
        self._check(self.lib.xvcgpu_set_stream(self.h, hip_stream))

    def record(self, fn):
        """Record the xvcgpu calls made by fn() into a replayable handle."""
        self._check(self.lib.xvcgpu_record_begin(self.h))
        try:
            fn()
        finally:
            h = _vp()
            st = self.lib.xvcgpu_record_end(self.h, C.byref(h))
        self._check(st)
        return h

    def replay(self, recording):
        self._check(self.lib.xvcgpu_replay(self.h, recording))

    def recording_destroy(self, recording):
        self.lib.xvcgpu_recording_destroy(recording)

    def use_priority_stream(self, high):
        self._check(self.lib.xvcgpu_use_priority_stream(self.h, 1 if high else 0))

    def set_short_kernel_priority(self, on=True):
        self._check(self.lib.xvcgpu_set_short_kernel_priority(self.h, int(on)))

    def wait_for(self, other):
        """Work queued on this context from now on starts after everything
        already queued on `other`."""
        self._check(self.lib.xvcgpu_wait_for(self.h, other.h))

    def stream_ptr(self):
        f = self.lib.xvcgpu_get_stream
        f.restype, f.argtypes = C.c_void_p, [_vp]
        return f(self.h) or 0

    def set_rdoq_four_lane_only(self, on):
        """xvcgpu_quant_rdo_set_four_lane_only: the batches hold no block of the general class."""
        self._check(self.lib.xvcgpu_quant_rdo_set_four_lane_only(self.h, 1 if on else 0))

    def set_rdoq_prove_zero(self, mode):
        """The all-zero proof ahead of the RDO quantiser's walk: 0 never, 1 always,
        -1 by batch size (xvcgpu_quant_rdo_set_prove_zero; same results either way)."""
        self._check(self.lib.xvcgpu_quant_rdo_set_prove_zero(self.h, int(mode)))

    def use_own_stream(self):
        self._check(self.lib.xvcgpu_use_own_stream(self.h))

    def timer_begin(self):
        self._check(self.lib.xvcgpu_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float(0)
        self._check(self.lib.xvcgpu_timer_end(self.h, C.byref(ms)))
        return ms.value

    def timer_mark(self, slot):
        self._check(self.lib.xvcgpu_timer_mark(self.h, slot))

    def timer_between(self, slot_a, slot_b):
        ms = C.c_float(0)
        self._check(self.lib.xvcgpu_timer_between(self.h, slot_a, slot_b, C.byref(ms)))
        return ms.value

    def picture(self, w, h, bd=10):
        return Picture(self, w, h, bd)

    def buffer(self, arr):
        return DeviceBuffer.from_array(self, arr)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def h2d(self, dev_ptr, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            self._check(self.lib.xvcgpu_memcpy_h2d(self.h, dev_ptr, arr.ctypes.data,
                                                   arr.nbytes))

    # ---- device-pointer level calls (bench) ----
    def pad_border(self, pic):
        self._check(self.lib.xvcgpu_pad_border(self.h, pic.h_pic))

    def me_search_dev(self, orig, ref, flags, d_blocks, n, d_results, max_size=64):
        self._check(self.lib.xvcgpu_me_search_sized(self.h, orig.h_pic, ref.h_pic,
                                                    flags, d_blocks, n, d_results,
                                                    max_size))

    def mc_batch_dev(self, ref, pred, d_blocks, n):
        self._check(self.lib.xvcgpu_mc_batch(self.h, ref.h_pic, pred.h_pic,
                                             d_blocks, n))

    def mc_bipred_batch_dev(self, ref0, ref1, pred, d_blocks, n):
        self._check(self.lib.xvcgpu_mc_bipred_batch(
            self.h, ref0.h_pic, ref1.h_pic, pred.h_pic, d_blocks, n))

    def bipred_search_dev(self, orig, ref_other, ref_search, d_jobs, n, d_results,
                          max_size=64):
        self._check(self.lib.xvcgpu_bipred_search(
            self.h, orig.h_pic, ref_other.h_pic, ref_search.h_pic, d_jobs, n,
            d_results, max_size))

    def mc_from_me_dev(self, ref, pred, d_blocks, d_results, n):
        self._check(self.lib.xvcgpu_mc_from_me(self.h, ref.h_pic, pred.h_pic,
                                               d_blocks, d_results, n))

    def cu_info_from_me_dev(self, d_blocks, d_results, d_nnz, d_luma_idx, n, qp_y,
                            qp_c, ref_poc, d_cus):
        self._check(self.lib.xvcgpu_cu_info_from_me(
            self.h, d_blocks, d_results, d_nnz, d_luma_idx, n, qp_y, qp_c, ref_poc,
            d_cus))

    def recon_from_me_dev(self, orig, ref, rec, d_blocks, d_results, n, qp_y, qp_c,
                          ref_poc, d_nnz, d_cus, intra_pic=0):
        self._check(self.lib.xvcgpu_recon_from_me(
            self.h, orig.h_pic, ref.h_pic, rec.h_pic, d_blocks, d_results, n, qp_y,
            qp_c, intra_pic, ref_poc, d_nnz, d_cus))

    def recon_from_me_rdoq_dev(self, orig, ref, rec, d_blocks, d_results, n, qp_y, qp_c,
                               ref_poc, d_nnz, d_cus, d_contexts, d_params, tx_flags=0):
        self._check(self.lib.xvcgpu_recon_from_me_rdoq(
            self.h, orig.h_pic, ref.h_pic, rec.h_pic, d_blocks, d_results, n, qp_y,
            qp_c, tx_flags, ref_poc, d_nnz, d_cus, d_contexts, d_params))

    def residual_rdoq_batch_dev(self, orig, pred, rec, d_blocks, n, d_levels, d_offsets, d_nnz,
                                d_contexts, d_params):
        self._check(self.lib.xvcgpu_residual_rdoq_batch(
            self.h, orig.h_pic, pred.h_pic, rec.h_pic, d_blocks, n, d_levels, d_offsets,
            d_nnz, d_contexts, d_params))

    def residual_batch_dev(self, orig, pred, rec, d_blocks, n, d_levels=None,
                           d_offsets=None, d_nnz=None):
        self._check(self.lib.xvcgpu_residual_batch(
            self.h, orig.h_pic, pred.h_pic, rec.h_pic, d_blocks, n, d_levels,
            d_offsets, d_nnz))

    def deblock_dev(self, rec, d_cus, n_cus, d_map, map_stride, bipred=0,
                    beta=0, tc=0, sub=4):
        self._check(self.lib.xvcgpu_deblock(self.h, rec.h_pic, d_cus, n_cus, d_map,
                                            map_stride, bipred, beta, tc, sub))

    def deblock_rows_dev(self, rec, d_cus, n_cus, d_map, map_stride, pass_, y0, y1,
                         bipred=0, beta=0, tc=0, sub=4):
        self._check(self.lib.xvcgpu_deblock_rows(self.h, rec.h_pic, d_cus, n_cus,
                                                 d_map, map_stride, bipred, beta, tc,
                                                 sub, pass_, y0, y1))

    def deblock_pad_ssd_dev(self, src, dst, orig, d_cus, n_cus, d_map, map_stride, bipred=0,
                            beta=0, tc=0, shift_bd=8, d_ssd=None):
        """xvcgpu_deblock_pad_ssd: deblocking src -> dst, PadBorder, luma SSD parts
        against `orig` (None: no SSD) in one launch; CUs all >= 8x8."""
        self._check(self.lib.xvcgpu_deblock_pad_ssd(
            self.h, src.h_pic, dst.h_pic, orig.h_pic if orig is not None else None, d_cus,
            n_cus, d_map, map_stride, bipred, beta, tc, shift_bd, d_ssd))

    def picture_ssd_dev(self, a, b, comp, shift_bd, d_out, y_begin=0, y_end=1 << 30):
        self._check(self.lib.xvcgpu_picture_ssd_rows(self.h, a.h_pic, b.h_pic, comp,
                                                     shift_bd, y_begin, y_end, d_out))

    # ---- numpy conveniences (tests) ----
    def metric_batch(self, a, b, comp, cands, weight=1.0, strength=16):
        cands = np.ascontiguousarray(cands, CAND_DTYPE)
        dc = self.buffer(cands)
        do = self.alloc(8 * max(1, len(cands)))
        self._check(self.lib.xvcgpu_metric_batch(self.h, a.h_pic, b.h_pic, comp,
                                                 weight, strength, dc.ptr,
                                                 len(cands), do.ptr))
        out = do.to_array(np.uint64, len(cands))
        dc.free()
        do.free()
        return out

    def me_search(self, orig, ref, blocks, flags=ME_FULLPEL | ME_SUBPEL,
                  results=None, max_size=64):
        blocks = np.ascontiguousarray(blocks, ME_DTYPE)
        db = self.buffer(blocks)
        if results is None:
            results = np.zeros(len(blocks), MERES_DTYPE)
        dr = self.buffer(np.ascontiguousarray(results, MERES_DTYPE))
        self.me_search_dev(orig, ref, flags, db.ptr, len(blocks), dr.ptr, max_size)
        out = dr.to_array(MERES_DTYPE, len(blocks))
        db.free()
        dr.free()
        return out

    def mc_batch(self, ref, pred, blocks):
        blocks = np.ascontiguousarray(blocks, MC_DTYPE)
        db = self.buffer(blocks)
        self.mc_batch_dev(ref, pred, db.ptr, len(blocks))
        self.sync()
        db.free()

    def mc_metric_batch(self, orig, ref, cands, strength=16):
        cands = np.ascontiguousarray(cands, MCM_DTYPE)
        dc = self.buffer(cands)
        do = self.alloc(8 * max(1, len(cands)))
        self._check(self.lib.xvcgpu_mc_metric_batch(self.h, orig.h_pic, ref.h_pic,
                                                    strength, dc.ptr, len(cands), do.ptr))
        out = do.to_array(np.uint64, len(cands))
        dc.free()
        do.free()
        return out

    def mc_affine_batch(self, ref, pred, blocks):
        blocks = np.ascontiguousarray(blocks, MCAFF_DTYPE)
        db = self.buffer(blocks)
        self._check(self.lib.xvcgpu_mc_affine_batch(self.h, ref.h_pic, pred.h_pic, db.ptr,
                                                    len(blocks)))
        self.sync()
        db.free()

    def mc_bipred_batch(self, ref0, ref1, pred, blocks):
        blocks = np.ascontiguousarray(blocks, MCBI_DTYPE)
        db = self.buffer(blocks)
        self.mc_bipred_batch_dev(ref0, ref1, pred, db.ptr, len(blocks))
        self.sync()
        db.free()

    def bipred_search(self, orig, ref_other, ref_search, jobs):
        jobs = np.ascontiguousarray(jobs, BI_DTYPE)
        dj = self.buffer(jobs)
        dr = self.buffer(np.zeros(len(jobs), MERES_DTYPE))
        self.bipred_search_dev(orig, ref_other, ref_search, dj.ptr, len(jobs), dr.ptr)
        out = dr.to_array(MERES_DTYPE, len(jobs))
        dj.free()
        dr.free()
        return out

    def bipred_search_lic(self, orig, ref_other, ref_search, rec, jobs, neighbours):
        """xvcgpu_bipred_search_lic: jobs BI_DTYPE, neighbours LIC_DTYPE (one per job)."""
        jobs = np.ascontiguousarray(jobs, BI_DTYPE)
        nb = np.ascontiguousarray(neighbours, LIC_DTYPE)
        assert len(nb) == len(jobs)
        dj, dn = self.buffer(jobs), self.buffer(nb)
        dr = self.buffer(np.zeros(len(jobs), MERES_DTYPE))
        self._check(self.lib.xvcgpu_bipred_search_lic(
            self.h, orig.h_pic, ref_other.h_pic, ref_search.h_pic, rec.h_pic, dj.ptr, dn.ptr,
            len(jobs), dr.ptr, 64))
        out = dr.to_array(MERES_DTYPE, len(jobs))
        for b in (dj, dn, dr):
            b.free()
        return out

    @staticmethod
    def level_offsets(blocks):
        sizes = blocks["w"].astype(np.int64) * blocks["h"].astype(np.int64)
        off = np.zeros(len(blocks), np.uint32)
        if len(blocks) > 1:
            off[1:] = np.cumsum(sizes)[:-1]
        return off, int(sizes.sum())

    def residual_batch(self, orig, pred, rec, blocks):
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        off, total = self.level_offsets(blocks)
        db = self.buffer(blocks)
        dof = self.buffer(off)
        dl = self.alloc(2 * max(1, total))
        dn = self.alloc(4 * max(1, len(blocks)))
        self.residual_batch_dev(orig, pred, rec, db.ptr, len(blocks), dl.ptr,
                                dof.ptr, dn.ptr)
        levels = dl.to_array(np.int16, total)
        nnz = dn.to_array(np.int32, len(blocks))
        for b in (db, dof, dl, dn):
            b.free()
        return levels, off, nnz

    def residual_rdoq_batch(self, orig, pred, rec, blocks, contexts, params):
        """TransformAndReconstruct with RdoQuant::QuantRdo for the blocks flagged
        TXF_RDOQ.  contexts: RDOQ_CTX_DTYPE[]; params: RDOQ_PARAMS_DTYPE[n]."""
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        off, total = self.level_offsets(blocks)
        db, dof = self.buffer(blocks), self.buffer(off)
        dc = self.buffer(np.ascontiguousarray(contexts, RDOQ_CTX_DTYPE))
        dp = self.buffer(np.ascontiguousarray(params, RDOQ_PARAMS_DTYPE))
        dl = self.alloc(2 * max(1, total))
        dn = self.alloc(4 * max(1, len(blocks)))
        self._check(self.lib.xvcgpu_residual_rdoq_batch(
            self.h, orig.h_pic, pred.h_pic, rec.h_pic, db.ptr, len(blocks), dl.ptr, dof.ptr,
            dn.ptr, dc.ptr, dp.ptr))
        levels = dl.to_array(np.int16, total)
        nnz = dn.to_array(np.int32, len(blocks))
        for b in (db, dof, dl, dn, dc, dp):
            b.free()
        return levels, off, nnz

    def quant_rdo_batch(self, bitdepth, blocks, coeffs, off, contexts, params):
        """RdoQuant::QuantRdo on coefficient blocks (w*h int16 at off[i])."""
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        db = self.buffer(blocks)
        dof = self.buffer(np.ascontiguousarray(off, np.uint32))
        dcf = self.buffer(np.ascontiguousarray(coeffs, np.int16))
        dc = self.buffer(np.ascontiguousarray(contexts, RDOQ_CTX_DTYPE))
        dp = self.buffer(np.ascontiguousarray(params, RDOQ_PARAMS_DTYPE))
        dl = self.alloc(2 * max(1, len(coeffs)))
        dn = self.alloc(4 * max(1, len(blocks)))
        self._check(self.lib.xvcgpu_quant_rdo_batch(
            self.h, bitdepth, db.ptr, len(blocks), dcf.ptr, dof.ptr, len(coeffs), dl.ptr,
            dn.ptr, dc.ptr, dp.ptr))
        try:
            self.sync()      # (reports a broken xvcgpu_quant_rdo_set_four_lane_only promise)
            levels = dl.to_array(np.int16, len(coeffs))
            nnz = dn.to_array(np.int32, len(blocks))
        finally:
            for b in (db, dof, dcf, dl, dn, dc, dp):
                b.free()
        return levels, nnz

    def fwd_transform_batch(self, orig, pred, blocks):
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        off, total = self.level_offsets(blocks)
        db = self.buffer(blocks)
        dof = self.buffer(off)
        dl = self.alloc(2 * max(1, total))
        self._check(self.lib.xvcgpu_fwd_transform_batch(
            self.h, orig.h_pic, pred.h_pic, db.ptr, len(blocks), dl.ptr, dof.ptr))
        coeffs = dl.to_array(np.int16, total)
        for b in (db, dof, dl):
            b.free()
        return coeffs, off

    def inv_transform_batch(self, pred, rec, blocks, levels, off, nnz):
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        db = self.buffer(blocks)
        dof = self.buffer(np.ascontiguousarray(off, np.uint32))
        dl = self.buffer(np.ascontiguousarray(levels, np.int16))
        dn = self.buffer(np.ascontiguousarray(nnz, np.int32))
        self._check(self.lib.xvcgpu_inv_transform_batch(
            self.h, pred.h_pic, rec.h_pic, db.ptr, len(blocks), dl.ptr, dof.ptr,
            dn.ptr))
        self.sync()
        for b in (db, dof, dl, dn):
            b.free()

    def inv_transform_dist_batch(self, orig, pred, rec, blocks, levels, off, nnz):
        """inv_transform_batch + the residual-domain SSD per block (uint64)."""
        blocks = np.ascontiguousarray(blocks, TX_DTYPE)
        db = self.buffer(blocks)
        dof = self.buffer(np.ascontiguousarray(off, np.uint32))
        dl = self.buffer(np.ascontiguousarray(levels, np.int16))
        dn = self.buffer(np.ascontiguousarray(nnz, np.int32))
        dd = self.alloc(8 * len(blocks))
        self._check(self.lib.xvcgpu_inv_transform_dist_batch(
            self.h, orig.h_pic, pred.h_pic, rec.h_pic, db.ptr, len(blocks), dl.ptr, dof.ptr,
            dn.ptr, dd.ptr))
        out = dd.to_array(np.uint64, len(blocks))
        for b in (db, dof, dl, dn, dd):
            b.free()
        return out

    def deblock(self, rec, cus, cu_map, bipred=0, beta=0, tc=0, sub=4):
        cus = np.ascontiguousarray(cus, CU_DTYPE)
        cu_map = np.ascontiguousarray(cu_map, np.int32)
        dc = self.buffer(cus)
        dm = self.buffer(cu_map)
        self.deblock_dev(rec, dc.ptr, len(cus), dm.ptr, cu_map.shape[1], bipred,
                         beta, tc, sub)
        self.sync()
        dc.free()
        dm.free()

    def deblock_pad_ssd(self, src, dst, orig, cus, cu_map, bipred=0, beta=0, tc=0, shift_bd=8):
        cus = np.ascontiguousarray(cus, CU_DTYPE)
        cu_map = np.ascontiguousarray(cu_map, np.int32)
        dc, dm, do = self.buffer(cus), self.buffer(cu_map), self.alloc(16)
        self.deblock_pad_ssd_dev(src, dst, orig, dc.ptr, len(cus), dm.ptr, cu_map.shape[1],
                                 bipred, beta, tc, shift_bd, do.ptr if orig is not None else None)
        self.sync()
        out = do.to_array(np.uint64, 2)
        for b in (dc, dm, do):
            b.free()
        return (int(out[0]), int(out[1])) if orig is not None else None

    # ---- whole-picture passes around the hot path ----
    def picture_import(self, pic, data, in_w, in_h, in_bd):
        """data: packed planar 4:2:0 bytes (Y,U,V) as the application holds
        them; staged in device memory, converted on the device."""
        src = np.frombuffer(data, np.uint8)
        assert len(src) == in_w * in_h * 3 // 2 * (2 if in_bd > 8 else 1)
        d = self.buffer(src)
        self._check(self.lib.xvcgpu_picture_import(self.h, pic.h_pic, d.ptr, in_w, in_h,
                                                   in_bd))
        self.sync()
        d.free()

    def picture_export(self, pic, disp_w, disp_h, out_bd, dither=False):
        n = disp_w * disp_h * 3 // 2 * (2 if out_bd > 8 else 1)
        d = self.alloc(n)
        self._check(self.lib.xvcgpu_picture_export(self.h, pic.h_pic, d.ptr, disp_w,
                                                   disp_h, out_bd, int(dither)))
        out = d.to_array(np.uint8, n).tobytes()
        d.free()
        return out

    def picture_crc(self, pic, mode=0):
        d = self.alloc(8)
        self._check(self.lib.xvcgpu_picture_crc(self.h, pic.h_pic, mode, d.ptr))
        out = d.to_array(np.uint8, 8)[:6 if mode else 2].tobytes()
        d.free()
        return out

    def variance_map(self, pic, ctu_size=64):
        """-> (16x16-block variances [rows, cols], per-CTU statistic [rows, cols])"""
        bw, bh = (pic.w + 15) // 16, (pic.h + 15) // 16
        cw, ch = -(-pic.w // ctu_size), -(-pic.h // ctu_size)
        dv, dc = self.alloc(8 * bw * bh), self.alloc(8 * cw * ch)
        self._check(self.lib.xvcgpu_variance_map(self.h, pic.h_pic, dv.ptr, ctu_size,
                                                 dc.ptr))
        v = dv.to_array(np.uint64, bw * bh).reshape(bh, bw)
        c = dc.to_array(np.uint64, cw * ch).reshape(ch, cw)
        dv.free()
        dc.free()
        return v, c

    def histogram_distance(self, a, b):
        d = self.alloc(8)
        self._check(self.lib.xvcgpu_histogram_distance(self.h, a.h_pic, b.h_pic, d.ptr))
        out = int(d.to_array(np.int64, 1)[0])
        d.free()
        return out

    def mc_lic_batch(self, ref, rec, pred, jobs):
        jobs = np.ascontiguousarray(jobs, LIC_DTYPE)
        d = self.buffer(jobs)
        self._check(self.lib.xvcgpu_mc_lic_batch(self.h, ref.h_pic, rec.h_pic, pred.h_pic,
                                                 d.ptr, len(jobs)))
        self.sync()
        d.free()

    def tx_eval_batch(self, jobs, alts):
        """CompressAndEvalTransform's fold per job -> TXE_RESULT_DTYPE array."""
        jobs = np.ascontiguousarray(jobs, TXE_JOB_DTYPE)
        alts = np.ascontiguousarray(alts, TXE_ALT_DTYPE)
        dj, da = self.buffer(jobs), self.buffer(alts if len(alts) else np.zeros(1, TXE_ALT_DTYPE))
        do = self.alloc(TXE_RESULT_DTYPE.itemsize * max(1, len(jobs)))
        self._check(self.lib.xvcgpu_tx_eval_batch(self.h, dj.ptr, len(jobs), da.ptr, do.ptr))
        out = do.to_array(TXE_RESULT_DTYPE, len(jobs))
        for b in (dj, da, do):
            b.free()
        return out

    def root_cbf_batch(self, jobs):
        """CompressAndEvalCbf's tail per CU -> ROOT_CBF_RESULT_DTYPE array."""
        jobs = np.ascontiguousarray(jobs, ROOT_CBF_JOB_DTYPE)
        dj = self.buffer(jobs)
        do = self.alloc(ROOT_CBF_RESULT_DTYPE.itemsize * max(1, len(jobs)))
        self._check(self.lib.xvcgpu_root_cbf_batch(self.h, dj.ptr, len(jobs), do.ptr))
        out = do.to_array(ROOT_CBF_RESULT_DTYPE, len(jobs))
        dj.free()
        do.free()
        return out

    def inter_pred_batch(self, refs, rec, pred, jobs):
        """InterPrediction::MotionCompensation per (CU, component) job - uni / bi,
        affine, local illumination compensation (reads `rec` around the CU) -
        from the reference pictures refs[job.ref[list]] into `pred`."""
        jobs = np.ascontiguousarray(jobs, INTER_DTYPE)
        arr = (_vp * max(1, len(refs)))(*[r.h_pic for r in refs])
        d = self.buffer(jobs)
        self._check(self.lib.xvcgpu_inter_pred_batch(self.h, arr, len(refs), rec.h_pic,
                                                     pred.h_pic, d.ptr, len(jobs)))
        d.free()

    def inter_pred_batch_to(self, refs, rec, scratch, jobs, dst):
        """xvcgpu_inter_pred_batch_to: job i's block goes to `scratch` at the luma
        position dst[i] (POS_DTYPE) instead of the CU's own."""
        jobs = np.ascontiguousarray(jobs, INTER_DTYPE)
        dst = np.ascontiguousarray(dst, POS_DTYPE)
        assert len(dst) == len(jobs)
        arr = (_vp * max(1, len(refs)))(*[r.h_pic for r in refs])
        d, dd = self.buffer(jobs), self.buffer(dst)
        self._check(self.lib.xvcgpu_inter_pred_batch_to(self.h, arr, len(refs), rec.h_pic,
                                                        scratch.h_pic, d.ptr, dd.ptr, len(jobs)))
        d.free()
        dd.free()

    def copy_blocks(self, src, dst, blocks):
        blocks = np.ascontiguousarray(blocks, COPY_BLOCK_DTYPE)
        d = self.buffer(blocks)
        self._check(self.lib.xvcgpu_copy_blocks(self.h, src.h_pic, dst.h_pic, d.ptr, len(blocks)))
        d.free()

    def affine_me_batch(self, orig, ref, blocks, ref_other=None):
        """InterSearch::MotionEstAffine per block -> AFFINE_ME_RESULT_DTYPE array
        (ref_other: the other list's reference for AFFINE_ME_BIPRED jobs)"""
        blocks = np.ascontiguousarray(blocks, AFFINE_ME_DTYPE)
        assert all(int(v) in (16, 32, 64) for v in np.concatenate([blocks["w"], blocks["h"]]))
        d = self.buffer(blocks)
        do = self.alloc(AFFINE_ME_RESULT_DTYPE.itemsize * max(1, len(blocks)))
        assert ref_other is not None or not (blocks["flags"] & AFFINE_ME_BIPRED).any()
        self._check(self.lib.xvcgpu_affine_me_batch(
            self.h, orig.h_pic, ref.h_pic, ref_other.h_pic if ref_other else None, d.ptr,
            len(blocks), do.ptr))
        out = do.to_array(AFFINE_ME_RESULT_DTYPE, len(blocks))
        d.free()
        do.free()
        return out

    # ---- one SearchMotion step into all reference pictures of a CU (xvcgpu_*_refs) ----
    def _ref_array(self, refs):
        return (C.c_void_p * len(refs))(*[r.h_pic for r in refs])

    def me_search_refs(self, orig, refs, blocks, slots, block_class, flags=ME_FULLPEL | ME_SUBPEL,
                       results=None):
        """job i searches refs[slots[i]] (255: no job) -> MERES_DTYPE array"""
        blocks = np.ascontiguousarray(blocks, ME_DTYPE)
        slots = np.ascontiguousarray(slots, np.uint8)
        assert len(slots) == len(blocks)
        db, ds = self.buffer(blocks), self.buffer(slots)
        if results is None:
            results = np.zeros(len(blocks), MERES_DTYPE)
        dr = self.buffer(np.ascontiguousarray(results, MERES_DTYPE))
        self._check(self.lib.xvcgpu_me_search_refs(self.h, orig.h_pic, self._ref_array(refs),
                                                   len(refs), flags, db.ptr, ds.ptr, len(blocks),
                                                   dr.ptr, block_class))
        out = dr.to_array(MERES_DTYPE, len(blocks))
        for b in (db, ds, dr):
            b.free()
        return out

    def bipred_search_refs(self, orig, refs, jobs, slots, block_class, results=None):
        """slots[i] = (searched picture, the other list's picture) of job i"""
        jobs = np.ascontiguousarray(jobs, BI_DTYPE)
        slots = np.ascontiguousarray(slots, np.uint8).reshape(-1, 2)
        assert len(slots) == len(jobs)
        dj, ds = self.buffer(jobs), self.buffer(slots)
        if results is None:
            results = np.zeros(len(jobs), MERES_DTYPE)
        dr = self.buffer(np.ascontiguousarray(results, MERES_DTYPE))
        self._check(self.lib.xvcgpu_bipred_search_refs(self.h, orig.h_pic, self._ref_array(refs),
                                                       len(refs), dj.ptr, ds.ptr, len(jobs), dr.ptr,
                                                       block_class))
        out = dr.to_array(MERES_DTYPE, len(jobs))
        for b in (dj, ds, dr):
            b.free()
        return out

    def mc_metric_batch_refs(self, orig, refs, cands, slots, strength=16):
        cands = np.ascontiguousarray(cands, MCM_DTYPE)
        slots = np.ascontiguousarray(slots, np.uint8)
        assert len(slots) == len(cands)
        dc, ds = self.buffer(cands), self.buffer(slots)
        do = self.buffer(np.full(max(1, len(cands)), 0xffffffffffffffff, np.uint64))
        self._check(self.lib.xvcgpu_mc_metric_batch_refs(self.h, orig.h_pic, self._ref_array(refs),
                                                         len(refs), strength, dc.ptr, ds.ptr,
                                                         len(cands), do.ptr))
        out = do.to_array(np.uint64, len(cands))
        for b in (dc, ds, do):
            b.free()
        return out

    def affine_me_batch_refs(self, orig, refs, blocks, slots, cu_height):
        blocks = np.ascontiguousarray(blocks, AFFINE_ME_DTYPE)
        slots = np.ascontiguousarray(slots, np.uint8).reshape(-1, 2)
        assert len(slots) == len(blocks)
        d, ds = self.buffer(blocks), self.buffer(slots)
        do = self.buffer(np.zeros(max(1, len(blocks)), AFFINE_ME_RESULT_DTYPE))
        self._check(self.lib.xvcgpu_affine_me_batch_refs(self.h, orig.h_pic, self._ref_array(refs),
                                                         len(refs), d.ptr, ds.ptr, len(blocks),
                                                         do.ptr, cu_height))
        out = do.to_array(AFFINE_ME_RESULT_DTYPE, len(blocks))
        for b in (d, ds, do):
            b.free()
        return out

    # ---- intra prediction ----
    def intra_pred_batch(self, rec, pred, jobs):
        jobs = np.ascontiguousarray(jobs, INTRA_DTYPE)
        d = self.buffer(jobs)
        self._check(self.lib.xvcgpu_intra_pred_batch(self.h, rec.h_pic, pred.h_pic, d.ptr,
                                                     len(jobs)))
        self.sync()
        d.free()

    def intra_satd_batch(self, orig, rec, jobs):
        """-> uint32 [len(jobs), 67]: SATD of every luma mode's prediction"""
        jobs = np.ascontiguousarray(jobs, INTRA_DTYPE)
        d = self.buffer(jobs)
        do = self.alloc(4 * INTRA_NUM_MODES * max(1, len(jobs)))
        ms = int(max(jobs["w"].max(), jobs["h"].max())) if len(jobs) else 4
        self._check(self.lib.xvcgpu_intra_satd_batch(self.h, orig.h_pic, rec.h_pic, d.ptr,
                                                     len(jobs), do.ptr, ms))
        out = do.to_array(np.uint32, INTRA_NUM_MODES * len(jobs)).reshape(-1, INTRA_NUM_MODES)
        d.free()
        do.free()
        return out

    def picture_ssd(self, a, b, comp=0, shift_bd=8):
        do = self.alloc(16)
        self.picture_ssd_dev(a, b, comp, shift_bd, do.ptr)
        out = do.to_array(np.uint64, 2)
        do.free()
        return int(out[0]), int(out[1])
