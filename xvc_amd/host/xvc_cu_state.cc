// xvc_cu_state.cc -- see xvc_cu_state.h.
#include "xvc_cu_state.h"

#include <chrono>
#include <cstring>
#include <utility>
#include <vector>

namespace {
const uint32_t kEntropyBits[128] = {
#include "../csrc/entropy_bits.inc"
};
const uint8_t kTransIdxLps[64] = {XVC_TRANS_IDX_LPS_LIST};
const xvc_bits_tables kTables = {kEntropyBits, kTransIdxLps};
}  // namespace

extern "C" {

void xvc_host_inter_pred_bits(const xvcgpu_inter_contexts *snapshots, const int32_t *ictx_index,
                              const xvc_inter_syntax *cands, int n, uint32_t *bits) {
  for (int i = 0; i < n; i++)
    bits[i] = xvc_inter_pred_bits(&snapshots[ictx_index ? ictx_index[i] : i], &cands[i], &kTables);
}

void xvc_host_next_state_table(int lps, uint8_t *out) {
  for (int s = 0; s < 128; s++)
    out[s] = xvc_ctx_next(static_cast<uint8_t>(s), lps ? !(s & 1) : (s & 1), kTransIdxLps);
}

const uint32_t *xvc_host_entropy_bits_table(void) { return kEntropyBits; }

}  // extern "C"

// ---- the serial walk ---------------------------------------------------------------
namespace {

struct Walk {
  xvcgpu_ctx *ctx;
  const xvc_cs_tables &t;
  xvc_cs_stats *st;
  int read_levels;
  xvcgpu_status err;
  Walk(xvcgpu_ctx *c, const xvc_cs_tables &tab, xvc_cs_stats *s, int rl)
      : ctx(c), t(tab), st(s), read_levels(rl), err(XVCGPU_OK) {}

  bool Ok(xvcgpu_status s) {
    st->api_calls++;
    if (s != XVCGPU_OK && err == XVCGPU_OK) err = s;
    return s == XVCGPU_OK;
  }
  // queue a device -> host copy of n records; the wait is ReadBack()
  template <typename T>
  void Fetch(T *host, const T *dev, size_t first, size_t n) {
    if (n && err == XVCGPU_OK) {
      const xvcgpu_status s = xvcgpu_memcpy_d2h_async(ctx, host + first, dev + first, n * sizeof(T));
      if (s != XVCGPU_OK) err = s;
    }
  }
  void ReadBack() {
    if (err != XVCGPU_OK) return;
    const xvcgpu_status s = xvcgpu_sync(ctx);
    if (s != XVCGPU_OK) err = s;
    st->round_trips++;
  }

  // jobs [first, first + n) that name the same reference slot(s) go out as one batch
  template <typename F>
  void ByRef(const int8_t *ref, int stride, int first, int n, F launch) {
    int a = first;
    while (a < first + n) {
      int b = a + 1;
      while (b < first + n && !std::memcmp(ref + (size_t)b * stride, ref + (size_t)a * stride, stride)) b++;
      launch(a, b - a, ref + (size_t)a * stride);
      a = b;
    }
  }

  // LIC states: the reconstruction of that moment around the CU (a live encoder's own
  // reconstruction picture holds it already)
  void Stage(const xvc_cs_state &s) {
    if (s.nb_count) Ok(xvcgpu_copy_blocks(ctx, t.nb, t.rec, t.d_nb_copy + s.nb_first, s.nb_count));
  }
  const xvcgpu_picture *Rec(const xvc_cs_state &s) const {
    return (s.flags & XVC_CS_STATE_LIC) ? t.rec : t.orig;
  }

  void Motion(const xvc_cs_state &s) {
    const int max_size = s.w > s.h ? s.w : s.h;
    if (s.me_count) {            // SearchRefIdx, uni-directional: one TZ + sub-pel search per picture
      ByRef(t.me_ref, 1, s.me_first, s.me_count, [&](int a, int n, const int8_t *r) {
        // (LIC states: the AC-only metrics, announced per call)
        Ok(xvcgpu_me_search_sized(ctx, t.orig, t.refs[r[0]],
                                  XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL |
                                      ((s.flags & XVC_CS_STATE_LIC) ? XVCGPU_ME_LIC_JOBS : 0),
                                  t.d_me + a, n, t.d_me_res + a, max_size));
      });
      Fetch(t.h_me_res, t.d_me_res, s.me_first, s.me_count);
      ReadBack();                // the fold over lists and pictures reads every result
    }
    if (s.bi_count) {            // SearchBiIterative: needs the uni-directional winners
      ByRef(t.bi_ref, 2, s.bi_first, s.bi_count, [&](int a, int n, const int8_t *r) {
        if (s.flags & XVC_CS_STATE_LIC)
          Ok(xvcgpu_bipred_search_lic(ctx, t.orig, t.refs[r[1]], t.refs[r[0]], t.rec, t.d_bi + a,
                                      t.d_bi_lic + a, n, t.d_bi_res + a, max_size));
        else
          Ok(xvcgpu_bipred_search(ctx, t.orig, t.refs[r[1]], t.refs[r[0]], t.d_bi + a, n,
                                  t.d_bi_res + a, max_size));
      });
      Fetch(t.h_bi_res, t.d_bi_res, s.bi_first, s.bi_count);
      ReadBack();
    }
    if (s.aff_uni_count) {       // second SearchMotion, affine: bootstrapped from the plain result
      ByRef(t.aff_ref, 2, s.aff_first, s.aff_uni_count, [&](int a, int n, const int8_t *r) {
        Ok(xvcgpu_affine_me_batch(ctx, t.orig, t.refs[r[0]], t.refs[r[1]], t.d_aff + a, n,
                                  t.d_aff_res + a));
      });
      Fetch(t.h_aff_res, t.d_aff_res, s.aff_first, s.aff_uni_count);
      ReadBack();
    }
    if (s.aff_bi_count) {
      const int f = s.aff_first + s.aff_uni_count;
      ByRef(t.aff_ref, 2, f, s.aff_bi_count, [&](int a, int n, const int8_t *r) {
        Ok(xvcgpu_affine_me_batch(ctx, t.orig, t.refs[r[0]], t.refs[r[1]], t.d_aff + a, n,
                                  t.d_aff_res + a));
      });
      Fetch(t.h_aff_res, t.d_aff_res, f, s.aff_bi_count);
      ReadBack();
    }
  }

  void MergeRank(const xvc_cs_state &s) {
    const size_t m = (size_t)s.merge * 5;
    Ok(xvcgpu_copy_blocks(ctx, t.orig, t.s_orig, t.d_mg_copy + m, 5));
    Ok(xvcgpu_inter_pred_batch_to(ctx, t.refs, t.n_refs, Rec(s), t.s_pred, t.d_mg_inter + m,
                                  t.d_mg_dst + m, 5));
    Ok(xvcgpu_metric_batch(ctx, t.s_orig, t.s_pred, 0, 1.0, 16, t.d_mg_cands + m, 5,
                           t.d_mg_dist + m));
    Fetch(t.h_mg_dist, t.d_mg_dist, m, 5);
    ReadBack();                  // the stable sort and the cut decide which candidates are evaluated
  }

  // calls [first, first + n) of evaluation e: residual pipeline + distortions
  void Calls(const xvc_cs_state &s, int first, int n, const int comp_count[3]) {
    if (!n) return;
    const size_t e = (size_t)s.ev;
    Ok(xvcgpu_copy_blocks(ctx, t.s_pred, t.s_pred, t.d_call_copy_pred + first, n));
    Ok(xvcgpu_residual_rdoq_batch(ctx, t.s_orig, t.s_pred, t.s_rec, t.d_call_tx + first, n,
                                  t.d_levels, t.d_call_off + first, t.d_nnz + first,
                                  t.d_contexts + t.ev_ctx[e], t.d_call_prm + first));
    int a = first;
    for (int c = 0; c < 3; c++) {
      if (!comp_count[c]) continue;
      Ok(xvcgpu_metric_batch(ctx, t.s_orig, t.s_rec, c, t.ev_weight[3 * e + c], 16,
                             t.d_call_cand + a, comp_count[c], t.d_call_dist + a));
      a += comp_count[c];
    }
    Fetch(t.h_nnz, t.d_nnz, first, n);
    Fetch(t.h_call_dist, t.d_call_dist, first, n);
  }

  void Eval(const xvc_cs_state &s) {
    const size_t e = (size_t)s.ev;
    const int n0 = s.call_pass0, n1 = s.call_pass1;
    // the originals beside the scratch predictions: slot 0 and the pass-0 slots
    Ok(xvcgpu_copy_blocks(ctx, t.orig, t.s_orig, t.d_copy_orig + s.copy_first, 3 + n0));
    Ok(xvcgpu_inter_pred_batch_to(ctx, t.refs, t.n_refs, Rec(s), t.s_pred, t.d_ev_inter + 3 * e,
                                  t.d_ev_dst + 3 * e, 3));
    for (int c = 0; c < 3; c++)   // cbf-zero distortion: the prediction against the original
      Ok(xvcgpu_metric_batch(ctx, t.s_orig, t.s_pred, c, t.ev_weight[3 * e + c], 16,
                             t.d_ev_dz + 3 * e + c, 1, t.d_ev_dz_dist + 3 * e + c));
    Calls(s, s.call_first, n0, s.comp_count);
    Fetch(t.h_ev_dz_dist, t.d_ev_dz_dist, 3 * e, 3);
    if (read_levels) Fetch(t.h_levels, t.d_levels, (size_t)s.level_first, LevelCount(s, 0));
    ReadBack();                  // bits of every alternative, the folds, the gate of the second pass
    if (n1) {
      const int cc[3] = {n1, 0, 0};
      Ok(xvcgpu_copy_blocks(ctx, t.orig, t.s_orig, t.d_copy_orig + s.copy_first + 3 + n0, n1));
      Calls(s, s.call_first + n0, n1, cc);
      if (read_levels)
        Fetch(t.h_levels, t.d_levels, (size_t)s.level_first + LevelCount(s, 0), LevelCount(s, 1));
      ReadBack();
    }
  }
  // CompressIntra: the SATD of the 67 luma modes, a wait (the host sorts with the mode
  // bits and keeps a few), then every PredictAndTransform alternative one behind the other
  // at the CU's own place - prediction from the staged reconstruction, transform + RDOQ +
  // reconstruction, distortion - with a wait wherever the reference reads a mode's cost
  void Intra(const xvc_cs_state &s) {
    const int max_size = s.w > s.h ? s.w : s.h;
    if (s.in_satd >= 0) {
      Ok(xvcgpu_intra_satd_batch(ctx, t.orig, t.rec, t.d_in_satd_jobs + s.in_satd, 1,
                                 t.d_in_satd + (size_t)67 * s.in_satd, max_size));
      Fetch(t.h_in_satd, t.d_in_satd, (size_t)67 * s.in_satd, 67);
      ReadBack();
    }
    int fetched = s.in_first;
    for (int c = s.in_first; c < s.in_first + s.in_count; c++) {
      if (t.in_stage[2 * c + 1])
        Ok(xvcgpu_copy_blocks(ctx, t.nb, t.rec, t.d_nb_copy + t.in_stage[2 * c], t.in_stage[2 * c + 1]));
      Ok(xvcgpu_intra_pred_batch(ctx, t.rec, t.ipred, t.d_in_pred + c, 1));
      Ok(xvcgpu_residual_rdoq_batch(ctx, t.orig, t.ipred, t.irec, t.d_in_tx + c, 1, t.d_in_levels,
                                    t.d_in_off + c, t.d_in_nnz + c, t.d_in_contexts + t.in_ctx[c],
                                    t.d_in_prm + c));
      Ok(xvcgpu_metric_batch(ctx, t.orig, t.irec, t.in_comp[c], t.in_weight[c], 16, t.d_in_cand + c, 1,
                             t.d_in_dist + c));
      if (t.in_wait[c]) {
        Fetch(t.h_in_nnz, t.d_in_nnz, fetched, c + 1 - fetched);
        Fetch(t.h_in_dist, t.d_in_dist, fetched, c + 1 - fetched);
        if (read_levels) {
          const size_t a = t.in_off_h[fetched], b = t.in_off_h[c + 1];   // (consecutive)
          Fetch(t.h_in_levels, t.d_in_levels, a, b - a);
        }
        ReadBack();
        fetched = c + 1;
      }
    }
  }
  // levels of a state's pass: the luma selections of pass 1 are w * h each
  size_t LevelCount(const xvc_cs_state &s, int pass) const {
    const size_t p1 = (size_t)s.call_pass1 * s.w * s.h;
    return pass ? p1 : (size_t)s.level_count - p1;
  }
};

double Now() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int xvc_host_cu_state_run_serial(xvcgpu_ctx *ctx, const xvc_cs_tables *t,
                                            const xvc_cs_state *states, int first, int n,
                                            int read_levels, xvc_cs_stats *stats) {
  if (!ctx || !t || !states || !stats || n < 0) return XVCGPU_INVALID_ARGUMENT;
  std::memset(stats, 0, sizeof(*stats));
  Walk w(ctx, *t, stats, read_levels);
  xvcgpu_status s0 = xvcgpu_sync(ctx);
  if (s0 != XVCGPU_OK) return s0;
  const double t0 = Now();
  for (int i = first; i < first + n && w.err == XVCGPU_OK; i++) {
    const xvc_cs_state &s = states[i];
    if (!s.supported) {
      stats->skipped++;
      continue;
    }
    const double a = Now();
    w.Stage(s);
    switch (s.kind) {
      case XVC_CS_MERGE_RANK: w.MergeRank(s); break;
      case XVC_CS_EVAL: w.Eval(s); break;
      case XVC_CS_INTER: w.Motion(s); w.Eval(s); break;
      case XVC_CS_MOTION: w.Motion(s); break;
      case XVC_CS_INTRA: w.Intra(s); break;
      default: return XVCGPU_INVALID_ARGUMENT;
    }
    stats->seconds_by_kind[s.kind] += Now() - a;
    stats->states_by_kind[s.kind]++;
    stats->states++;
  }
  stats->seconds = Now() - t0;
  return w.err;
}

extern "C" void xvc_host_cs_sizes(int32_t out[7]) {
  out[0] = static_cast<int32_t>(sizeof(xvc_cs_state));
  out[1] = static_cast<int32_t>(sizeof(xvc_cs_tables));
  out[2] = static_cast<int32_t>(sizeof(xvc_cs_stats));
  out[3] = static_cast<int32_t>(sizeof(xvcgpu_cs_pass));
  out[4] = static_cast<int32_t>(sizeof(xvcgpu_cs_result));
  out[5] = static_cast<int32_t>(sizeof(xvc_cs_op));
  out[6] = static_cast<int32_t>(sizeof(xvc_cs_env));
}

// ---- the chained form: a state (or all states of a CU position) as ONE enqueue ---------
// What is data-dependent between the steps of a state is composed on the device by the
// folds of csrc/k_cu_state.h; what the host issues per state is therefore a fixed
// sequence of launches on arrays it filled before - recorded here as a small program
// (tests/rd_serial.py writes it from the state table; an encoder would emit the same
// ops as it walks its CU tree).  No op looks at a result: the only wait is the SYNC that
// ends a chain.
namespace {
// one op of a chain program (SYNC is the caller's)
xvcgpu_status IssueOp(xvcgpu_ctx *ctx, const xvc_cs_env *env, const xvc_cs_op &o) {
  const xvcgpu_picture *pics[8] = {env->orig, env->s_orig, env->s_pred, env->s_rec,
                                   env->nb,   env->rec,    env->ipred,  env->irec};
  void *const *p = reinterpret_cast<void *const *>(o.p);
  // r0 / r1 name a reference slot or one of the four pictures above, by opcode
  const bool ref_slots = o.opcode == XVC_OP_MC_METRIC || o.opcode == XVC_OP_ME ||
                         o.opcode == XVC_OP_BI || o.opcode == XVC_OP_AFFINE ||
                         o.opcode == XVC_OP_BI_LIC;
  const bool pic_index = o.opcode == XVC_OP_METRIC || o.opcode == XVC_OP_COPY ||
                         o.opcode == XVC_OP_INTER_PRED;
  if (ref_slots && (o.r0 < 0 || o.r0 >= env->n_refs ||
                    (o.opcode != XVC_OP_MC_METRIC && o.opcode != XVC_OP_ME &&
                     (o.r1 < 0 || o.r1 >= env->n_refs))))
    return XVCGPU_INVALID_ARGUMENT;
  if (pic_index && (o.r1 < 0 || o.r1 > 7 || !pics[o.r1] ||
                    (o.opcode != XVC_OP_INTER_PRED && (o.r0 < 0 || o.r0 > 7 || !pics[o.r0]))))
    return XVCGPU_INVALID_ARGUMENT;
  if ((o.opcode == XVC_OP_INTRA_SATD || o.opcode == XVC_OP_INTRA_PRED ||
       o.opcode == XVC_OP_RESIDUAL_INTRA) && (!env->rec || !env->ipred || !env->irec))
    return XVCGPU_INVALID_ARGUMENT;
  if ((o.opcode == XVC_OP_BI_LIC || (o.opcode == XVC_OP_INTER_PRED && o.r0 == 1)) && !env->rec)
    return XVCGPU_INVALID_ARGUMENT;
  switch (o.opcode) {
    case XVC_OP_MC_METRIC:
      return xvcgpu_mc_metric_batch(ctx, env->orig, env->refs[o.r0], 16,
                                    static_cast<const xvcgpu_mc_metric_cand *>(p[0]), o.n,
                                    static_cast<uint64_t *>(p[1]));
    case XVC_OP_METRIC:
      return xvcgpu_metric_batch(ctx, pics[o.r0], pics[o.r1], o.i0, o.f, 16,
                                 static_cast<const xvcgpu_metric_cand *>(p[0]), o.n,
                                 static_cast<uint64_t *>(p[1]));
    case XVC_OP_ME:
      return xvcgpu_me_search_sized(ctx, env->orig, env->refs[o.r0],   // r1 = 1: LIC jobs
                                    XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL |
                                        (o.r1 == 1 ? XVCGPU_ME_LIC_JOBS : 0),
                                    static_cast<const xvcgpu_me_block *>(p[0]), o.n,
                                    static_cast<xvcgpu_me_result *>(p[1]), o.i0);
    case XVC_OP_BI:
      return xvcgpu_bipred_search(ctx, env->orig, env->refs[o.r1], env->refs[o.r0],
                                  static_cast<const xvcgpu_bi_block *>(p[0]), o.n,
                                  static_cast<xvcgpu_me_result *>(p[1]), o.i0);
    case XVC_OP_BI_LIC:
      return xvcgpu_bipred_search_lic(ctx, env->orig, env->refs[o.r1], env->refs[o.r0], env->rec,
                                      static_cast<const xvcgpu_bi_block *>(p[0]),
                                      static_cast<const xvcgpu_mc_lic_block *>(p[2]), o.n,
                                      static_cast<xvcgpu_me_result *>(p[1]), o.i0);
    case XVC_OP_AFFINE:
      return xvcgpu_affine_me_batch(ctx, env->orig, env->refs[o.r0], env->refs[o.r1],
                                    static_cast<const xvcgpu_affine_me_block *>(p[0]), o.n,
                                    static_cast<xvcgpu_affine_me_result *>(p[1]));
    case XVC_OP_COPY:
      return xvcgpu_copy_blocks(ctx, pics[o.r0], const_cast<xvcgpu_picture *>(pics[o.r1]),
                                static_cast<const xvcgpu_copy_block *>(p[0]), o.n);
    case XVC_OP_INTER_PRED:
      // r0 = 1: XVC_INTER_LIC jobs among them - the neighbours from the reconstruction
      return xvcgpu_inter_pred_batch_to(ctx, env->refs, env->n_refs,
                                        o.r0 == 1 ? env->rec : env->orig,
                                        const_cast<xvcgpu_picture *>(pics[o.r1]),
                                        static_cast<const xvcgpu_inter_block *>(p[0]),
                                        static_cast<const xvcgpu_block_pos *>(p[1]), o.n);
    case XVC_OP_RESIDUAL:
      if (p[5])   // originals from the picture itself, the prediction from its one slot
        return xvcgpu_residual_rdoq_batch_at(ctx, env->orig, env->s_pred, env->s_rec,
                                             static_cast<const xvcgpu_tx_block *>(p[0]), o.n,
                                             env->d_levels, static_cast<const uint32_t *>(p[1]),
                                             static_cast<int32_t *>(p[2]),
                                             static_cast<const xvcgpu_rdoq_contexts *>(p[3]),
                                             static_cast<const xvcgpu_rdoq_params *>(p[4]),
                                             static_cast<const xvcgpu_block_pos *>(p[5]), 16,
                                             // p[6] / p[7]: the evaluation's candidates and their
                                             // distortions, priced by the same launch (r0 = the
                                             // prediction-against-original candidates in front)
                                             static_cast<const xvcgpu_eval_cand *>(p[6]), o.r0,
                                             static_cast<uint64_t *>(p[7]));
      return xvcgpu_residual_rdoq_batch(ctx, env->s_orig, env->s_pred, env->s_rec,
                                        static_cast<const xvcgpu_tx_block *>(p[0]), o.n,
                                        env->d_levels, static_cast<const uint32_t *>(p[1]),
                                        static_cast<int32_t *>(p[2]),
                                        static_cast<const xvcgpu_rdoq_contexts *>(p[3]),
                                        static_cast<const xvcgpu_rdoq_params *>(p[4]));
    case XVC_OP_START_FOLD:
      return xvcgpu_cs_start_fold(ctx, static_cast<const xvcgpu_cs_pass *>(p[0]), o.i0, o.n,
                                  static_cast<const uint64_t *>(p[1]),
                                  static_cast<xvcgpu_me_block *>(p[2]),
                                  static_cast<const xvcgpu_me_result *>(p[3]),
                                  static_cast<xvcgpu_affine_me_block *>(p[4]), env->d_results,
                                  env->pic_w, env->pic_h);
    case XVC_OP_UNI_FOLD:
      return xvcgpu_cs_uni_fold(ctx, static_cast<const xvcgpu_cs_pass *>(p[0]), o.i0, o.n,
                                static_cast<const xvcgpu_me_result *>(p[1]),
                                static_cast<const xvcgpu_affine_me_result *>(p[2]),
                                env->d_results, static_cast<xvcgpu_bi_block *>(p[3]),
                                static_cast<xvcgpu_affine_me_block *>(p[4]));
    case XVC_OP_BI_FOLD:
      return xvcgpu_cs_bi_fold(ctx, static_cast<const xvcgpu_cs_pass *>(p[0]), o.i0, o.n,
                               static_cast<const xvcgpu_me_result *>(p[1]),
                               static_cast<const xvcgpu_affine_me_result *>(p[2]),
                               env->d_results, static_cast<xvcgpu_inter_block *>(p[3]));
    case XVC_OP_EVAL_DIST:   // r0 = 1: the candidates name their original's place in env->orig
      return xvcgpu_eval_dist_batch(ctx, o.r0 == 1 ? env->orig : env->s_orig, env->s_pred,
                                    env->s_rec, 16,
                                    static_cast<const xvcgpu_eval_cand *>(p[0]), o.n,
                                    static_cast<uint64_t *>(p[1]));
    case XVC_OP_FETCH:
      return xvcgpu_memcpy_d2h_async(ctx, p[1], p[0], static_cast<size_t>(o.n));
    case XVC_OP_MC_METRIC_REFS:
      return xvcgpu_mc_metric_batch_refs(ctx, env->orig, env->refs, env->n_refs, 16,
                                         static_cast<const xvcgpu_mc_metric_cand *>(p[0]),
                                         static_cast<const uint8_t *>(p[2]), o.n,
                                         static_cast<uint64_t *>(p[1]));
    case XVC_OP_ME_REFS:
      return xvcgpu_me_search_refs(ctx, env->orig, env->refs, env->n_refs,
                                   XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL,
                                   static_cast<const xvcgpu_me_block *>(p[0]),
                                   static_cast<const uint8_t *>(p[2]), o.n,
                                   static_cast<xvcgpu_me_result *>(p[1]), o.i0);
    case XVC_OP_BI_REFS:
      return xvcgpu_bipred_search_refs(ctx, env->orig, env->refs, env->n_refs,
                                       static_cast<const xvcgpu_bi_block *>(p[0]),
                                       static_cast<const uint8_t *>(p[2]), o.n,
                                       static_cast<xvcgpu_me_result *>(p[1]), o.i0);
    case XVC_OP_AFFINE_REFS:
      return xvcgpu_affine_me_batch_refs(ctx, env->orig, env->refs, env->n_refs,
                                         static_cast<const xvcgpu_affine_me_block *>(p[0]),
                                         static_cast<const uint8_t *>(p[2]), o.n,
                                         static_cast<xvcgpu_affine_me_result *>(p[1]), o.i0);
    case XVC_OP_INTRA_SATD:
      return xvcgpu_intra_satd_batch(ctx, env->orig, env->rec,
                                     static_cast<const xvcgpu_intra_block *>(p[0]), o.n,
                                     static_cast<uint32_t *>(p[1]), o.i0);
    case XVC_OP_INTRA_PRED:
      return xvcgpu_intra_pred_batch(ctx, env->rec, env->ipred,
                                     static_cast<const xvcgpu_intra_block *>(p[0]), o.n);
    case XVC_OP_RESIDUAL_INTRA:
      return xvcgpu_residual_rdoq_batch(ctx, env->orig, env->ipred, env->irec,
                                        static_cast<const xvcgpu_tx_block *>(p[0]), o.n,
                                        env->d_in_levels, static_cast<const uint32_t *>(p[1]),
                                        static_cast<int32_t *>(p[2]),
                                        static_cast<const xvcgpu_rdoq_contexts *>(p[3]),
                                        static_cast<const xvcgpu_rdoq_params *>(p[4]));
    case XVC_OP_MERGE_FOLD:
      return xvcgpu_cs_merge_fold(ctx, static_cast<const xvcgpu_cs_merge *>(p[0]), o.i0, o.n,
                                  static_cast<const uint64_t *>(p[1]),
                                  static_cast<const xvcgpu_inter_block *>(p[2]),
                                  static_cast<xvcgpu_cs_merge_result *>(p[3]),
                                  static_cast<xvcgpu_inter_block *>(p[4]));
    default:
      return XVCGPU_INVALID_ARGUMENT;
  }
}
}  // namespace

extern "C" int xvc_host_cs_run_program(xvcgpu_ctx *ctx, const xvc_cs_env *env,
                                       const xvc_cs_op *ops, int64_t n_ops, xvc_cs_stats *stats) {
  if (!ctx || !env || !ops || !stats || n_ops < 0) return XVCGPU_INVALID_ARGUMENT;
  std::memset(stats, 0, sizeof(*stats));
  xvcgpu_status st = xvcgpu_sync(ctx);
  if (st != XVCGPU_OK) return st;
  const double t0 = Now();
  double chain_t0 = t0;
  for (int64_t i = 0; i < n_ops && st == XVCGPU_OK; i++) {
    const xvc_cs_op &o = ops[i];
    if (o.opcode == XVC_OP_SYNC) {   // the end of a chain: i0 = states it held, r0 = kind
      st = xvcgpu_sync(ctx);
      stats->round_trips++;
      stats->states += o.i0;
      const double now = Now();
      const int k = o.r0 >= 0 && o.r0 < XVC_CS_KINDS ? o.r0 : 0;
      stats->seconds_by_kind[k] += now - chain_t0;
      stats->states_by_kind[k] += o.i0;
      chain_t0 = now;
      continue;
    }
    st = IssueOp(ctx, env, o);
    if (o.opcode != XVC_OP_FETCH) stats->api_calls++;   // (a copy is not a launch)
  }
  stats->seconds = Now() - t0;
  return st;
}

// k independent programs (k pictures in flight, each with its own context = stream,
// arrays and scratch) driven by ONE thread: a chain's next state sequence is issued
// while the other chains' run, and the thread comes back to wait for a chain only after
// it has issued work for all the others.  (One thread per chain contends for the
// runtime's submission path: four threads reach 2.5x of one, not 4x.)
extern "C" int xvc_host_cs_run_programs_interleaved(int k, xvcgpu_ctx *const *ctxs,
                                                    const xvc_cs_env *const *envs,
                                                    const xvc_cs_op *const *ops,
                                                    const int64_t *n_ops, xvc_cs_stats *stats) {
  if (k < 1 || k > 64 || !ctxs || !envs || !ops || !n_ops || !stats)
    return XVCGPU_INVALID_ARGUMENT;
  std::memset(stats, 0, sizeof(*stats));
  int64_t at[64];
  int64_t waiting[64];               // states of the chain whose results are awaited, or -1
  for (int c = 0; c < k; c++) {
    at[c] = 0;
    waiting[c] = -1;
    const xvcgpu_status st = xvcgpu_sync(ctxs[c]);
    if (st != XVCGPU_OK) return st;
  }
  const double t0 = Now();
  int live = k;
  xvcgpu_status st = XVCGPU_OK;
  while (live > 0 && st == XVCGPU_OK) {
    live = 0;
    for (int c = 0; c < k && st == XVCGPU_OK; c++) {
      if (waiting[c] >= 0) {         // issued a round ago: its results
        st = xvcgpu_sync(ctxs[c]);
        stats->round_trips++;
        stats->states += waiting[c];
        waiting[c] = -1;
      }
      while (at[c] < n_ops[c] && st == XVCGPU_OK) {
        const xvc_cs_op &o = ops[c][at[c]++];
        if (o.opcode == XVC_OP_SYNC) {
          waiting[c] = o.i0;
          break;
        }
        st = IssueOp(ctxs[c], envs[c], o);
        if (o.opcode != XVC_OP_FETCH) stats->api_calls++;
      }
      if (waiting[c] >= 0 || at[c] < n_ops[c]) live++;
    }
  }
  // (an error in one chain: the others' work is waited for, the device is left idle)
  if (st != XVCGPU_OK)
    for (int c = 0; c < k; c++) xvcgpu_sync(ctxs[c]);
  stats->seconds = Now() - t0;
  return st;
}

// ---- the engine: many chains, one launch per step kind -----------------------------------
// k recorded programs on ONE context (one stream).  A round takes every chain's next step,
// groups the steps by kind (and kernel instance) and issues one xvcgpu_cs_segs_launch per
// group - the chains' jobs side by side in the grid's y - so the launch path carries
// (kinds per round) launches for k chain steps instead of k.  A chain's steps stay in order
// because a round issues at most one launch-step per chain and the stream is in order; a
// chain that reaches its SYNC records an event and sits out until the event has passed
// (the others keep the device busy).  Steps without a batched form are issued as they are.
namespace {
int SegKindOf(const xvc_cs_op &o) {
  switch (o.opcode) {
    case XVC_OP_MC_METRIC_REFS: return XVC_CS_SEG_MC_METRIC_REFS;
    case XVC_OP_START_FOLD: return XVC_CS_SEG_START_FOLD;
    case XVC_OP_UNI_FOLD: return XVC_CS_SEG_UNI_FOLD;
    case XVC_OP_BI_FOLD: return XVC_CS_SEG_BI_FOLD;
    case XVC_OP_MERGE_FOLD: return XVC_CS_SEG_MERGE_FOLD;
    case XVC_OP_ME_REFS: return XVC_CS_SEG_ME_REFS;
    case XVC_OP_BI_REFS: return XVC_CS_SEG_BI_REFS;
    case XVC_OP_AFFINE_REFS: return XVC_CS_SEG_AFFINE_REFS;
    // into s_pred; LIC jobs (r0 = 1: neighbours from the chain's reconstruction) as they are
    case XVC_OP_INTER_PRED: return (o.r1 == 2 && o.r0 != 1) ? XVC_CS_SEG_INTER_PRED : -1;
    case XVC_OP_RESIDUAL: return o.p[5] ? XVC_CS_SEG_RESIDUAL_AT : -1;
    case XVC_OP_EVAL_DIST: return o.r0 == 1 ? XVC_CS_SEG_EVAL_DIST : -1;
    default: return -1;
  }
}
}  // namespace

// how long a kind's launch lasts, roughly (to deal the groups of a round over the streams)
static int SegWeight(int kind, int key) {
  switch (kind) {
    // (microseconds per launch, profiles/r06_cu_state_engine_kernel_stats.csv)
    case XVC_CS_SEG_BI_REFS: return key >= 64 ? 37 : (key >= 32 ? 24 : 21);
    case XVC_CS_SEG_AFFINE_REFS: return key >= 64 ? 34 : (key >= 32 ? 36 : 42);
    case XVC_CS_SEG_RESIDUAL_AT: return 34;
    case XVC_CS_SEG_ME_REFS: return key >= 64 ? 34 : (key >= 32 ? 32 : 14);
    default: return 9;
  }
}

extern "C" int xvc_host_cs_run_programs_engine(xvcgpu_ctx *const *ctxs, int n_ctx, int k,
                                               const xvc_cs_env *const *envs,
                                               const xvc_cs_op *const *ops, const int64_t *n_ops,
                                               xvc_cs_stats *stats) {
  if (!ctxs || n_ctx < 1 || n_ctx > 8 || k < 1 || k > 256 || !envs || !ops || !n_ops || !stats)
    return XVCGPU_INVALID_ARGUMENT;
  for (int i = 0; i < n_ctx; i++)
    if (!ctxs[i]) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_ctx *ctx = ctxs[0];
  std::memset(stats, 0, sizeof(*stats));
  std::vector<xvcgpu_cs_env *> denv(k, nullptr);
  std::vector<xvcgpu_event *> round_ev(n_ctx, nullptr);
  xvcgpu_status st = XVCGPU_OK;
  for (int c = 0; c < k && st == XVCGPU_OK; c++) {
    const xvc_cs_env *e = envs[c];
    st = xvcgpu_cs_env_create(ctx, e->orig, e->refs, e->n_refs, e->s_orig, e->s_pred, e->s_rec,
                              e->d_levels, e->d_results, &denv[c]);
  }
  for (int i = 0; i < n_ctx && st == XVCGPU_OK; i++) {
    st = xvcgpu_event_create(ctxs[i], &round_ev[i]);
    if (st == XVCGPU_OK) st = xvcgpu_sync(ctxs[i]);
  }
  std::vector<int64_t> at(k, 0), wait_states(k, -1);
  // groups of a round: kind x kernel instance (i0 of the searches)
  struct Group {
    int kind, key;
    std::vector<xvcgpu_cs_seg> segs;
  };
  std::vector<Group> groups;
  std::vector<int> order, syncs;
  std::vector<xvcgpu_cs_seg> fetches;
  std::vector<char> fetched(k, 0);
  const double t0 = Now();
  int live = k;
  bool used_last[8] = {false}, used_now[8] = {false}, met[8] = {false};
  // the chains that reach their SYNC in one round share its event (one record, one query)
  const int kSyncEvents = 64;
  std::vector<xvcgpu_event *> sync_ev(kSyncEvents, nullptr);
  std::vector<int> sync_waiters(kSyncEvents, 0), qcache(kSyncEvents, -1), ev_of(k, -1);
  for (int i = 0; i < kSyncEvents && st == XVCGPU_OK; i++) st = xvcgpu_event_create(ctx, &sync_ev[i]);
  int qcache_round = 0, sync_next = 0;
  std::vector<int> qcache_at(kSyncEvents, -1);
  auto Meet = [&](int t) -> xvcgpu_status {
    if (met[t]) return XVCGPU_OK;
    met[t] = true;
    for (int sidx = 0; sidx < n_ctx; sidx++)
      if (sidx != t && used_last[sidx]) {
        const xvcgpu_status r = xvcgpu_event_wait(ctxs[t], round_ev[sidx]);
        if (r != XVCGPU_OK) return r;
      }
    return XVCGPU_OK;
  };
  while (live > 0 && st == XVCGPU_OK) {
    live = 0;
    bool issued = false;
    int first_waiting = -1;
    for (Group &g : groups) g.segs.clear();
    fetches.clear();
    syncs.clear();
    // The streams meet between rounds: a chain's step of this round may run on another
    // stream than its step of the last one (a round's groups are dealt over the streams so
    // that a long search does not hold the others up)
    // (a stream waits for the streams that had work in the last round when it gets its first
    // launch of this one: Meet below)
    for (int t = 0; t < n_ctx; t++) {
      used_last[t] = used_now[t];
      used_now[t] = false;
      met[t] = false;
    }
    qcache_round++;
    for (int c = 0; c < k && st == XVCGPU_OK; c++) {
      if (wait_states[c] >= 0) {           // at its SYNC: has the event passed?
        int done = 0;
        const int e = ev_of[c];
        if (qcache_at[e] != qcache_round) {
          st = xvcgpu_event_query(sync_ev[e], &qcache[e]);
          if (st != XVCGPU_OK) break;
          qcache_at[e] = qcache_round;
        }
        done = qcache[e];
        if (!done) {
          if (first_waiting < 0) first_waiting = c;
          live++;
          continue;
        }
        stats->round_trips++;
        stats->states += wait_states[c];
        wait_states[c] = -1;
        sync_waiters[e]--;
      }
      while (at[c] < n_ops[c] && st == XVCGPU_OK) {
        const xvc_cs_op &o = ops[c][at[c]];
        if (o.opcode == XVC_OP_SYNC) {     // (behind the chain's read-backs: stream 0, below)
          syncs.push_back(c);
          wait_states[c] = o.i0;
          at[c]++;
          issued = true;
          break;
        }
        if (o.opcode == XVC_OP_FETCH) {
          // behind the chain's last launch (an earlier round).  The round's read-backs are
          // one launch of their own, on stream 0 in front of the SYNC events
          if ((o.n & 3) || ((o.p[0] | o.p[1]) & 3)) {
            st = IssueOp(ctx, envs[c], o);
          } else {
            xvcgpu_cs_seg sg;
            std::memset(&sg, 0, sizeof(sg));
            sg.n = o.n;
            sg.p[0] = o.p[0];
            sg.p[1] = o.p[1];
            fetches.push_back(sg);
          }
          at[c]++;
          fetched[c] = true;
          // the read-backs of a round go out on stream 0 behind the chains' walk: a launch
          // step that follows one (no SYNC between) must not be issued in the same round -
          // it could overwrite what the read-back has not copied yet.  It waits for the next
          // round, whose launches are behind this round's events.
          if (at[c] < n_ops[c] && ops[c][at[c]].opcode != XVC_OP_FETCH &&
              ops[c][at[c]].opcode != XVC_OP_SYNC) {
            issued = true;
            break;
          }
          continue;
        }
        const int kind = SegKindOf(o);
        if (kind < 0) {                    // no batched form (a LIC state's steps): as it is,
          const int t = c % n_ctx;         // behind the last round's launches of every stream
          st = Meet(t);
          used_now[t] = true;
          if (st == XVCGPU_OK) st = IssueOp(ctxs[t], envs[c], o);
          stats->api_calls++;
        } else {
          const int key = (kind == XVC_CS_SEG_ME_REFS || kind == XVC_CS_SEG_BI_REFS ||
                           kind == XVC_CS_SEG_AFFINE_REFS) ? o.i0 : 0;
          Group *grp = nullptr;
          for (Group &g : groups)
            if (g.kind == kind && g.key == key) grp = &g;
          if (!grp) {
            groups.push_back(Group());
            grp = &groups.back();
            grp->kind = kind;
            grp->key = key;
          }
          xvcgpu_cs_seg sg;
          sg.n = o.n;
          sg.i0 = o.i0;
          sg.r0 = o.r0;
          sg.r1 = o.r1;
          for (int q = 0; q < 8; q++) sg.p[q] = o.p[q];
          sg.env = denv[c];
          grp->segs.push_back(sg);
        }
        at[c]++;
        issued = true;
        break;                             // one launch step per chain and round
      }
      if (wait_states[c] >= 0 || at[c] < n_ops[c]) live++;
    }
    // the round's read-backs, then the events of the chains that wait for them
    if ((!fetches.empty() || !syncs.empty()) && st == XVCGPU_OK) st = Meet(0);
    if (!fetches.empty() && st == XVCGPU_OK) {
      st = xvcgpu_cs_segs_launch(ctx, XVC_CS_SEG_FETCH, fetches.data(), static_cast<int>(fetches.size()));
      stats->api_calls++;
      used_now[0] = true;
    }
    if (!syncs.empty() && st == XVCGPU_OK) {
      int e = -1;
      for (int tries = 0; tries < kSyncEvents && e < 0; tries++) {
        const int cand = (sync_next + tries) % kSyncEvents;
        if (sync_waiters[cand] == 0) e = cand;
      }
      if (e < 0) {
        st = XVCGPU_DEVICE_ERROR;          // (k <= 256 chains can hold 64 events only if they
      } else {                             //  all wait, and then nothing reaches a SYNC)
        sync_next = (e + 1) % kSyncEvents;
        st = xvcgpu_event_record(ctx, sync_ev[e]);
        qcache_at[e] = -1;
        for (size_t i = 0; i < syncs.size(); i++) ev_of[syncs[i]] = e;
        sync_waiters[e] = static_cast<int>(syncs.size());
      }
    }
    // the round's groups, the long ones first, dealt over the streams (least loaded next)
    order.clear();
    for (size_t i = 0; i < groups.size(); i++)
      if (!groups[i].segs.empty()) order.push_back(static_cast<int>(i));
    for (size_t i = 1; i < order.size(); i++)
      for (size_t j2 = i; j2 > 0 && SegWeight(groups[order[j2]].kind, groups[order[j2]].key) >
                                      SegWeight(groups[order[j2 - 1]].kind, groups[order[j2 - 1]].key);
           j2--)
        std::swap(order[j2], order[j2 - 1]);
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < order.size() && st == XVCGPU_OK; i++) {
      Group &g = groups[order[i]];
      int s_min = 0;
      for (int t = 1; t < n_ctx; t++)
        if (load[t] < load[s_min]) s_min = t;
      load[s_min] += SegWeight(g.kind, g.key);
      st = Meet(s_min);
      used_now[s_min] = true;
      if (st == XVCGPU_OK) st = xvcgpu_cs_segs_launch(ctxs[s_min], g.kind, g.segs.data(), static_cast<int>(g.segs.size()));
      stats->api_calls++;
    }
    if (n_ctx > 1)
      for (int t = 0; t < n_ctx && st == XVCGPU_OK; t++)
        if (used_now[t]) st = xvcgpu_event_record(ctxs[t], round_ev[t]);
    // every chain waits for the device: wait for one of them instead of spinning
    if (!issued && first_waiting >= 0 && st == XVCGPU_OK)
      st = xvcgpu_event_synchronize(sync_ev[ev_of[first_waiting]]);
  }
  for (int i = 0; i < n_ctx; i++) {
    const xvcgpu_status s2 = xvcgpu_sync(ctxs[i]);
    if (st == XVCGPU_OK) st = s2;
  }
  stats->seconds = Now() - t0;
  for (int c = 0; c < k; c++) {
    if (denv[c]) xvcgpu_cs_env_destroy(denv[c]);
  }
  for (int i = 0; i < n_ctx; i++)
    if (round_ev[i]) xvcgpu_event_destroy(round_ev[i]);
  for (int i = 0; i < kSyncEvents; i++)
    if (sync_ev[i]) xvcgpu_event_destroy(sync_ev[i]);
  return st;
}
