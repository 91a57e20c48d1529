// xvc_cu_state_builder.h -- the COMPOSER of the CU-state walk: from what the encoder's
// control code holds when it reaches a CU state to the records, job arrays and op programs
// xvc_cu_state.cc runs.
//
// Where this sits in the reference: CuEncoder::CompressInterPic / CompressMerge /
// CompressInter (xvc_enc_lib/cu_encoder.cc:431-541, :579-642) hold a CU, the picture's
// reference lists, the AMVP predictors per (list, picture) (InterPrediction::GetMvpList
// [Affine]), the merge candidates, lambda and the entropy coder's context states, and hand
// them to InterSearch::SearchMotion (inter_search.cc:199-259) / SearchMergeCandidates
// (:165-197) / CompressAndEvalCbf (:261-365).  A device-side encoder hands the SAME things
// to xvc_gpu::CuStateBuilder, one state after the other in its own issue order, and gets
//   * the xvcgpu_cs_pass records of the states' SearchMotion passes (plain, affine, LIC),
//   * the work arrays the device folds compose the searches' jobs in (uni-directional,
//     refinement slots, affine), the EvalStartMvp candidates, the per-job reference slots
//     of the *_refs entry points,
//   * the merge rankings' fold records and evaluation slots, the evaluations' distortion
//     candidates,
//   * the op program of a stretch of states in any of the walk's forms (a chain per
//     state, per visit of a CU position, the chains a live encoder could issue).
// Rounds 4 - 5 composed all of this in tests/rd_serial.py; that module now only turns the
// captured encode into the input records below (tests/test_cu_state_builder.py holds the
// C++ output against digests of what the Python composer produced).
#ifndef XVC_AMD_HOST_XVC_CU_STATE_BUILDER_H_
#define XVC_AMD_HOST_XVC_CU_STATE_BUILDER_H_

#include <cstdint>

#include "xvc_cu_state.h"

// ---- inputs ----------------------------------------------------------------------------
// One (list, picture) SearchRefIdx visits for a CU (inter_search.cc:456-578) with the two
// predictors GetMvpList / GetMvpListAffine gave for it ([predictor][corner][x, y]; a plain
// vector uses corner 0).
typedef struct xvc_csb_ref_entry {
  int8_t list, ref_idx;
  uint8_t reused;          // a list-1 picture that is also in list 0: its search is re-used (:536-542)
  uint8_t reserved;
  int32_t mvp[2][3][2];
} xvc_csb_ref_entry;

// One SearchMotion pass as the control code knows it before the search.
typedef struct xvc_csb_pass_in {
  int32_t first, n;        // its entries (n = 0: the pass is not run); n = pictures of both lists
  uint32_t lambda16;       // floor(65536 * sqrt(lambda))
  uint8_t fullpel;         // cu.GetFullpelMv()
  uint8_t reserved[3];
  xvcgpu_inter_contexts ictx;
} xvc_csb_pass_in;

// CompressInter's SearchMotion [+ the affine second pass] of one CU state.
typedef struct xvc_csb_motion {
  int32_t state;           // index in the state table
  int32_t nb;              // LIC: the CU's neighbour record, -1: none
  xvc_csb_pass_in plain, affine;
} xvc_csb_motion;

// Where the reconstruction above / left of a LIC CU lies (xvcgpu_mc_lic_block's fields).
typedef struct xvc_csb_neighbours {
  uint8_t has_above, has_left;
  int16_t above_x, above_y, left_x, left_y;
} xvc_csb_neighbours;

// One merge ranking (SearchMergeCandidates): the CU and sqrt(lambda); its five candidates'
// prediction jobs are rows 5 m .. 5 m + 4 of the merge job arrays.
typedef struct xvc_csb_merge {
  double lambda_sqrt;
  int16_t x, y;
  uint8_t w, h;
  uint8_t any_lic;         // one of the candidates uses LIC: the slots carry the neighbour fields
  uint8_t reserved;
  int32_t nb;              // neighbour record, -1
  int32_t state;           // the ranking's state
} xvc_csb_merge;

// One evaluation (CompressAndEvalCbf of a given motion): position, the three cbf-zero
// distortion candidates with their weights, and the TransformAndReconstruct calls
// [call_first, call_first + n_calls) with one distortion candidate + component each.
typedef struct xvc_csb_eval {
  int16_t x, y;
  int32_t state;
  int32_t merge_slot;      // the evaluation slot a merge fold fills for it, -1: its own jobs
  xvcgpu_metric_cand dz[3];
  double weight[3];
} xvc_csb_eval;

// The picture-wide tables (host memory; the builder keeps the pointers until Build returns).
typedef struct xvc_csb_picture {
  const xvc_cs_state *states;
  int32_t n_states;
  int32_t ref_poc[2][XVC_CS_MAX_REFS];     // the picture's reference lists
  int32_t n_ref[2];
  const int32_t *slot_pocs;                // picture slot -> POC (the refs array of the tables)
  int32_t n_slots;
  int32_t lic_folds;                       // LIC states folded on the device too
  const xvc_csb_motion *motions;           // the inter / motion states, in state order
  int32_t n_motions;
  const xvc_csb_ref_entry *entries;
  const xvc_csb_neighbours *nb;
  const xvcgpu_me_block *me_jobs;          // the serial form's job arrays (the searches'
  const int8_t *me_ref;                    // inputs as the reference had them) and the slot
  int32_t n_me;                            // of every job's picture
  const xvcgpu_affine_me_block *aff_jobs;
  const int8_t *aff_ref;                   // [job][searched, other]
  int32_t n_aff;
  const xvcgpu_inter_block *ev_inter;      // 3 per evaluation
  int32_t n_ev;
  const xvc_csb_merge *merges;
  int32_t n_merges;
  const xvc_csb_eval *evals;               // n_ev
  const int32_t *ev_ctx;                   // the evaluation's RDOQ context snapshot
  const xvcgpu_metric_cand *call_cand;     // per TransformAndReconstruct call
  const uint8_t *call_comp;
  const int32_t *call_ev;                  // the call's evaluation
  int32_t n_calls;
  const xvcgpu_metric_cand *mg_cands;      // 5 per ranking (the serial form's candidates)
} xvc_csb_picture;

// ---- outputs ---------------------------------------------------------------------------
enum {
  XVC_CSB_PASSES = 0, XVC_CSB_PASS_FIRST, XVC_CSB_PASS_COUNT, XVC_CSB_FOLDED,
  XVC_CSB_START_CANDS, XVC_CSB_START_SLOTS, XVC_CSB_AFF_START_INTER, XVC_CSB_AFF_START_DST,
  XVC_CSB_AFF_START_CANDS, XVC_CSB_AFF_START_COPY, XVC_CSB_ME_WORK, XVC_CSB_BI_LIC_WORK,
  XVC_CSB_AFF_WORK, XVC_CSB_AFF_WORK_SRC, XVC_CSB_ME_SLOTS, XVC_CSB_BI_SLOTS, XVC_CSB_AFF_SLOTS,
  XVC_CSB_EV_INTER_WORK, XVC_CSB_MG_FOLD, XVC_CSB_MG_SLOTS, XVC_CSB_MERGE_STATE,
  XVC_CSB_EV_CANDS, XVC_CSB_EV_CANDS_COPY, XVC_CSB_EDIST_FIRST, XVC_CSB_CALL_POS,
  XVC_CSB_MG_ECANDS, XVC_CSB_AFF_START_ECANDS, XVC_CSB_ARRAYS
};

// The addresses a program's ops refer to (device pointers unless marked host; what
// ChainedRun of tests/rd_serial.py allocates).
typedef struct xvc_csb_addrs {
  // tables of the serial form (xvc_cs_tables)
  uint64_t d_me, d_me_res, h_me_res, d_bi, d_bi_res, h_bi_res, d_bi_lic;
  uint64_t d_nb_copy, d_mg_copy, d_mg_inter, d_mg_dst, d_mg_cands;
  uint64_t d_ev_dst, d_copy_orig, d_call_copy_pred, d_call_tx, d_call_off, d_call_prm, d_contexts;
  uint64_t d_levels, h_levels;
  uint64_t d_in_satd_jobs, d_in_satd, h_in_satd, d_in_pred, d_in_tx, d_in_off, d_in_nnz, h_in_nnz;
  uint64_t d_in_contexts, d_in_prm, d_in_cand, d_in_dist, h_in_dist, d_in_levels, h_in_levels;
  // arrays of the chained form (uploads of the builder's outputs, result arrays)
  uint64_t passes, start_cands, start_slots, start_dist, aff_start_inter, aff_start_dst;
  uint64_t aff_start_cands, aff_start_copy, aff_start_ecands, me_work, me_res_c, me_slots;
  uint64_t aff_work, aff_res_c, aff_slots, bi_work, bi_res_c, bi_slots, bi_lic_work;
  uint64_t ev_inter_work, results, h_results, h_ev_inter_out;
  uint64_t mg_fold, mg_slots, mg_ecands, z_mg_dist, z_mg_res, z_mg_slots_out;
  uint64_t ev_cands, ev_cands_copy, call_pos, z_nnz, z_edist;
} xvc_csb_addrs;

// Per-call tables of the intra states the program reads (host).
typedef struct xvc_csb_intra {
  const int32_t *in_stage;   // [call][first, count] of its block copies
  const int32_t *in_ctx, *in_comp;
  const double *in_weight;
  const uint32_t *in_off;    // level offsets
  int32_t n_in;
  int64_t n_in_levels;
  const int8_t *bi_ref;      // the serial refinement jobs' slots [job][searched, other]
} xvc_csb_intra;

enum { XVC_CSB_BY_POSITION = 1, XVC_CSB_VERIFY = 2, XVC_CSB_REFS_FORM = 4, XVC_CSB_LIVE = 8,
       XVC_CSB_NO_COPIES = 16, XVC_CSB_FUSED_EVAL = 32, XVC_CSB_MERGE_FOLD = 64 };

#ifdef __cplusplus
extern "C" {
#endif
typedef struct xvc_csb xvc_csb;
// Builds every array from the picture's tables.  0: ok; < 0: an input the folds do not run
// (a pass with a forced zero L1 difference or more than one refinement iteration, more start
// candidates than scratch slots, entries that do not cover both lists) - *out is NULL then.
int xvc_host_csb_build(const xvc_csb_picture *pic, xvc_csb **out);
void xvc_host_csb_destroy(xvc_csb *b);
// Array `which` (XVC_CSB_*): its first byte and its length in BYTES.
const void *xvc_host_csb_array(const xvc_csb *b, int which, int64_t *bytes);
int32_t xvc_host_csb_n_start_dist(const xvc_csb *b);
int32_t xvc_host_csb_n_bi_slots(const xvc_csb *b);
int64_t xvc_host_csb_n_edist(const xvc_csb *b);
// The op program of the states [first, first + n) (flags: XVC_CSB_*); *n_ops its length.
// The returned array lives until the next call or the builder's destruction.
const xvc_cs_op *xvc_host_csb_program(xvc_csb *b, const xvc_csb_addrs *a, const xvc_csb_intra *in,
                                      int32_t first, int32_t n, int32_t flags, int64_t *n_ops);
#ifdef __cplusplus
}
#endif

#endif  // XVC_AMD_HOST_XVC_CU_STATE_BUILDER_H_
