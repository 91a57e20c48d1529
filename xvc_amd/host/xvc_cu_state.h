// xvc_cu_state.h -- one CU STATE of the encoder's RD search at a time, in the order
// the reference issues them.
//
// CuEncoder::CompressCu / CompressInterPic (xvc_enc_lib/cu_encoder.cc:123-273,
// :431-515) evaluate a CU position mode by mode; each mode is a CHAIN whose every
// step consumes the previous step's result:
//   CompressInter (inter_search.cc:74-98)
//     SearchMotion (:199-259): per list and picture EvalStartMvp -> TZ search +
//       sub-pel -> EvalFinalMvpIdx -> GetInterPredBits -> cost fold (SearchRefIdx
//       :456-578); SearchBiIterative (:392-433); the three-way choice (:247-257)
//     [second SearchMotion with the affine model, :85-93]
//     CompressAndEvalCbf (:261-365): motion compensation, per component the
//       TransformAndReconstruct alternatives, distortions
//   CompressMerge (cu_encoder.cc:598-...): SearchMergeCandidates (:165-197) then one
//     CompressAndEvalCbf per kept candidate
// and the next mode / the next CU starts from what this one decided (best cost,
// CABAC state, neighbouring vectors, reconstruction).  A bit-exact encoder can only
// present this work one state at a time; this layer is the device form for that
// regime: what a state costs when each step is a batch of one with a read-back
// wherever the reference reads a result (RunSerial: the honest baseline), and the
// same chain as ONE enqueue with the folds on the device (xvcgpu_cu_state_*,
// RunChained).  SURVEY 8d: "latency / launch bounds the per-CU batches; report it as
// such".
#ifndef XVC_AMD_HOST_XVC_CU_STATE_H_
#define XVC_AMD_HOST_XVC_CU_STATE_H_

#include <cstdint>

#include "xvc_inter_bits.h"
#include "xvcgpu.h"

// One CU state (tests/rd_serial.py builds the table from a captured encode; an
// encoder's control code would fill one record as it goes).  Ranges index the job
// arrays of xvc_cs_tables.
//   CompressIntra (cu_encoder.cc:518-541): DetermineSlowIntraModes (the SATD of the 67 luma
//     modes, intra_search.cc:188-305) -> the host sorts with the mode bits -> every kept
//     luma mode's and every chroma mode's PredictAndTransform (:61-82, :118-150)
enum { XVC_CS_MERGE_RANK = 0, XVC_CS_EVAL = 1, XVC_CS_INTER = 2, XVC_CS_MOTION = 3,
       XVC_CS_INTRA = 4, XVC_CS_KINDS = 5 };
typedef struct xvc_cs_state {
  int32_t kind;
  int16_t x, y;
  uint8_t w, h, flags, supported;
  int32_t me_first, me_count;               // uni-directional TZ + sub-pel searches
  int32_t bi_first, bi_count;               // bi-prediction refinement steps
  int32_t aff_first, aff_uni_count, aff_bi_count;
  int32_t merge;                            // merge ranking: index of the 5-candidate group
  int32_t ev;                               // evaluation: index into the per-evaluation arrays
  int32_t call_first, call_pass0, call_pass1;   // TransformAndReconstruct calls
  int32_t comp_count[3];                    // pass-0 calls per component
  int32_t copy_first;                       // originals: [3 (slot 0)] [pass 0] [pass 1]
  int32_t cand_first, cand_count, final_first, final_count;
  // XVC_CS_STATE_LIC: the block copies that stage the reconstruction of that moment above /
  // left of the CU (tables.nb -> tables.rec), issued in front of the state's jobs
  int32_t nb_first, nb_count;
  // XVC_CS_INTRA: the SATD pre-selection's job (-1: none) and the state's
  // TransformAndReconstruct calls (ranges of the in_* arrays)
  int32_t in_satd, in_first, in_count, in_reserved;
  int64_t level_first, level_count;
} xvc_cs_state;
// flags of a state: the whole-sample search (1) and local illumination compensation (2):
// cu.GetUseLic() - AC-only metrics in the searches, the other list's prediction compensated
// in the refinement (xvcgpu_bipred_search_lic), XVC_INTER_LIC prediction jobs
enum { XVC_CS_STATE_FULLPEL = 1, XVC_CS_STATE_LIC = 2 };

// Device job / result arrays of one picture's states, in issue order (all device
// memory unless marked host).  Scratch geometry: slot k of a state at luma x = 64 k of
// the three scratch pictures (originals, predictions, reconstructions).
typedef struct xvc_cs_tables {
  const xvcgpu_picture *orig;
  const xvcgpu_picture *const *refs;        // host array of handles
  int32_t n_refs;
  xvcgpu_picture *s_orig, *s_pred, *s_rec;  // scratch pictures (>= 512 x 64)
  // uni-directional searches
  const xvcgpu_me_block *d_me;
  xvcgpu_me_result *d_me_res;
  const int8_t *me_ref;                     // host: reference slot per job
  // bi-prediction steps
  const xvcgpu_bi_block *d_bi;
  xvcgpu_me_result *d_bi_res;
  const int8_t *bi_ref;                     // host: [job][searched, other]
  // affine searches
  const xvcgpu_affine_me_block *d_aff;
  xvcgpu_affine_me_result *d_aff_res;
  const int8_t *aff_ref;                    // host: [job][searched, other]
  // merge rankings (5 candidates each)
  const xvcgpu_inter_block *d_mg_inter;
  const xvcgpu_block_pos *d_mg_dst;
  const xvcgpu_copy_block *d_mg_copy;
  const xvcgpu_metric_cand *d_mg_cands;
  uint64_t *d_mg_dist;
  // evaluations
  xvcgpu_inter_block *d_ev_inter;           // 3 per evaluation (written by the chained form)
  const xvcgpu_block_pos *d_ev_dst;
  const xvcgpu_metric_cand *d_ev_dz;        // 3 per evaluation
  uint64_t *d_ev_dz_dist;
  const double *ev_weight;                  // host: 3 per evaluation (Qp::GetDistortionWeight)
  const int32_t *ev_ctx;                    // host: context snapshot per evaluation
  const xvcgpu_rdoq_contexts *d_contexts;
  const xvcgpu_copy_block *d_copy_orig;
  // calls
  const xvcgpu_tx_block *d_call_tx;
  const xvcgpu_rdoq_params *d_call_prm;
  const uint32_t *d_call_off;               // level offsets
  const xvcgpu_copy_block *d_call_copy_pred;
  const xvcgpu_metric_cand *d_call_cand;
  int16_t *d_levels;
  int32_t *d_nnz;
  uint64_t *d_call_dist;
  // host mirrors the read-backs land in (same indexing as the device result arrays)
  xvcgpu_me_result *h_me_res, *h_bi_res;
  xvcgpu_affine_me_result *h_aff_res;
  uint64_t *h_mg_dist, *h_ev_dz_dist, *h_call_dist;
  int32_t *h_nnz;
  int16_t *h_levels;
  // local illumination compensation: the chain's reconstruction picture (the references'
  // size; a live encoder's own reconstruction), the staging picture the captured neighbour
  // rows / columns are copied from, the copies, and per bi-prediction job its neighbours
  xvcgpu_picture *rec;
  const xvcgpu_picture *nb;
  const xvcgpu_copy_block *d_nb_copy;
  const xvcgpu_mc_lic_block *d_bi_lic;
  // intra states: prediction and reconstruction pictures (the references' size: an intra
  // block lies at its own place), the SATD jobs and their 67 distortions each, and per
  // TransformAndReconstruct call the prediction job, transform block, quantiser parameters,
  // level offset, distortion candidate; host: context snapshot, distortion weight,
  // component, [first, count] of the call's block copies (its reference samples when they
  // changed), whether the reference reads a cost behind the call
  xvcgpu_picture *ipred, *irec;
  const xvcgpu_intra_block *d_in_satd_jobs;
  uint32_t *d_in_satd, *h_in_satd;
  const xvcgpu_intra_block *d_in_pred;
  const xvcgpu_tx_block *d_in_tx;
  const xvcgpu_rdoq_params *d_in_prm;
  const uint32_t *d_in_off;
  const xvcgpu_metric_cand *d_in_cand;
  const xvcgpu_rdoq_contexts *d_in_contexts;
  const int32_t *in_ctx;
  const double *in_weight;
  const int32_t *in_comp, *in_stage, *in_wait;
  const uint32_t *in_off_h;                 // host: d_in_off and the total behind it
  int16_t *d_in_levels;
  int32_t *d_in_nnz;
  uint64_t *d_in_dist;
  int32_t *h_in_nnz;
  uint64_t *h_in_dist;
  int16_t *h_in_levels;
} xvc_cs_tables;

typedef struct xvc_cs_stats {
  double seconds;          // wall time of the walk
  int64_t states;          // states run (unsupported ones are skipped and not counted)
  int64_t skipped;
  int64_t api_calls;       // C-ABI entry points called
  int64_t round_trips;     // device -> host read-backs the walk waited for
  double seconds_by_kind[XVC_CS_KINDS];
  int64_t states_by_kind[XVC_CS_KINDS];
} xvc_cs_stats;

// The chained form's program: one op = one C-ABI call on arrays the host filled before
// (p[]: device pointers; FETCH: p[0] device -> p[1] host, n bytes).  r0 / r1: reference
// picture slots, or picture selectors 0 original, 1 scratch originals, 2 scratch
// predictions, 3 scratch reconstructions (METRIC, COPY, INTER_PRED).  The *_REFS ops are
// a step into all the CU's reference pictures as one launch (xvcgpu_*_refs): p[0] jobs,
// p[1] results, p[2] the jobs' slot bytes, i0 = block class / CU height.
// Ordering inside a chain: ops take effect in program order.  The engine
// (xvc_host_cs_run_engine) collects a round's FETCH ops into one copy launch behind the
// round's walk; a launch step that follows a FETCH without a SYNC between is held back
// to the next round, so it cannot overwrite an array the copy has not read yet.
enum {
  XVC_OP_MC_METRIC = 0, XVC_OP_METRIC, XVC_OP_ME, XVC_OP_BI, XVC_OP_AFFINE, XVC_OP_COPY,
  XVC_OP_INTER_PRED, XVC_OP_RESIDUAL, XVC_OP_START_FOLD, XVC_OP_UNI_FOLD, XVC_OP_BI_FOLD,
  XVC_OP_FETCH, XVC_OP_SYNC, XVC_OP_EVAL_DIST, XVC_OP_MC_METRIC_REFS, XVC_OP_ME_REFS,
  XVC_OP_BI_REFS, XVC_OP_AFFINE_REFS,
  XVC_OP_MERGE_FOLD,  // p: merges, distortions, candidates' jobs, results, evaluation slots; i0 = first
  // LIC states (picture selectors 4 = the neighbour staging picture, 5 = the chain's
  // reconstruction; INTER_PRED with r0 = 1 reads the neighbours from the reconstruction; ME
  // with r1 = 1 announces XVCGPU_ME_LIC_JOBS):
  XVC_OP_BI_LIC,      // xvcgpu_bipred_search_lic: r0 searched / r1 other slot, p: jobs, results, neighbours
  // intra states (picture selectors 6 = the intra prediction picture, 7 = the intra
  // reconstruction picture):
  XVC_OP_INTRA_SATD,  // xvcgpu_intra_satd_batch(orig, rec): p: jobs, distortions; i0 = block size
  XVC_OP_INTRA_PRED,  // xvcgpu_intra_pred_batch(rec -> ipred): p: jobs
  XVC_OP_RESIDUAL_INTRA  // xvcgpu_residual_rdoq_batch(orig, ipred, irec): p as RESIDUAL's first five
};
typedef struct xvc_cs_op {
  int32_t opcode, n, r0, r1, i0, reserved;
  double f;
  uint64_t p[8];
} xvc_cs_op;
typedef struct xvc_cs_env {
  const xvcgpu_picture *orig;
  const xvcgpu_picture *const *refs;
  int32_t n_refs, pic_w, pic_h, reserved;
  xvcgpu_picture *s_orig, *s_pred, *s_rec;
  int16_t *d_levels;
  xvcgpu_cs_result *d_results;
  xvcgpu_picture *rec;          // LIC / intra states only (may be NULL without them)
  const xvcgpu_picture *nb;
  xvcgpu_picture *ipred, *irec; // intra states only
  int16_t *d_in_levels;
} xvc_cs_env;

extern "C" {
int xvc_host_cs_run_program(xvcgpu_ctx *ctx, const xvc_cs_env *env, const xvc_cs_op *ops,
                            int64_t n_ops, xvc_cs_stats *stats);
// k chains (their own contexts, environments, programs) driven by one thread, a chain's
// next sequence issued while the others' run.
int xvc_host_cs_run_programs_interleaved(int k, xvcgpu_ctx *const *ctxs,
                                         const xvc_cs_env *const *envs,
                                         const xvc_cs_op *const *ops, const int64_t *n_ops,
                                         xvc_cs_stats *stats);
// k recorded programs through the engine: every round the chains' next steps grouped by
// kind, one xvcgpu_cs_segs_launch per kind, the round's groups dealt over the n_ctx streams
// (contexts on one device; they meet between rounds by events); k <= 256, n_ctx <= 8
int xvc_host_cs_run_programs_engine(xvcgpu_ctx *const *ctxs, int n_ctx, int k,
                                    const xvc_cs_env *const *envs,
                                    const xvc_cs_op *const *ops, const int64_t *n_ops,
                                    xvc_cs_stats *stats);

// The states [first, first + n) walked ONE AT A TIME with the batched entry points as
// they are: every step a batch of one CU, a read-back (copy + wait) wherever the
// reference reads a result before it can go on - after the uni-directional searches,
// after each bi-prediction / affine stage, after the first transform pass (the
// second one is gated on its cost, inter_search.cc:347-361) and after the last.
// read_levels: also bring the levels to the host (the entropy coder prices them).
int xvc_host_cu_state_run_serial(xvcgpu_ctx *ctx, const xvc_cs_tables *t,
                                 const xvc_cs_state *states, int first, int n, int read_levels,
                                 xvc_cs_stats *stats);

// sizeof of xvc_cs_state, xvc_cs_tables, xvc_cs_stats, xvcgpu_cs_pass, xvcgpu_cs_result,
// xvc_cs_op, xvc_cs_env: bindings check.
void xvc_host_cs_sizes(int32_t out[7]);

// InterSearch::GetInterPredBits with the encoder's default setting, for n motion
// candidates: candidate i is priced against snapshots[ictx_index[i]] (ictx_index
// NULL: snapshot i).  Host arithmetic (include/xvc_inter_bits.h).
void xvc_host_inter_pred_bits(const xvcgpu_inter_contexts *snapshots, const int32_t *ictx_index,
                              const xvc_inter_syntax *cands, int n, uint32_t *bits);
// The context state machine's transition as the product computes it (test hook):
// out[s] = next state byte after an MPS (lps == 0) / LPS bin, s = 0 .. 127.
void xvc_host_next_state_table(int lps, uint8_t *out);
const uint32_t *xvc_host_entropy_bits_table(void);
}

#endif  // XVC_AMD_HOST_XVC_CU_STATE_H_
