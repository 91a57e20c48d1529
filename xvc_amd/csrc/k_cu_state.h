// k_cu_state.h -- the folds BETWEEN the searches of one SearchMotion, on the device.
//
// InterSearch::SearchMotion (xvc_enc_lib/inter_search.cc:199-259) as the reference
// runs it for one CU is a chain of dependent steps; the expensive ones are the
// batched kernels of this library (k_me2.h, k_bipred.h, k_affine_me.h).  Between them
// the reference does a few dozen integer operations per candidate: EvalStartMvp's
// two-way choice (:966-997), EvalFinalMvpIdx (:999-1020), the vector difference
// (SetMvd :1022-1048), GetInterPredBits (:1082-1137, default setting: the candidate's
// syntax through a throw-away entropy coder - include/xvc_inter_bits.h), cost =
// dist + ((bits * lambda) >> 16), first strictly cheaper wins (:563-571), the
// choice of the list SearchBiIterative searches (:243-245, :406-407) and the
// three-way choice (:247-257).  Done on the host that is one read-back per step
// (xvc_host_cu_state_run_serial measures it); here each fold is a launch of one
// thread per pass that reads the previous step's results and WRITES THE NEXT STEP'S
// JOBS, so a CU state is enqueued once and read back once.
//
// One thread per pass does the arithmetic on purpose: ~200 dependent scalar operations
// on a handful of candidates - nothing to spread over lanes.  What does cost time is
// memory latency: read field by field from global memory the pass record (412 bytes)
// and the running result (448 bytes) are a chain of ~1 us round trips (13.7 / 11.9 us
// per launch, profiles/r04_cu_state_chained_kernel_stats.csv).  So a wave per pass
// copies both records into LDS in one round trip, lane 0 works there, and the wave
// writes the result back.
#ifndef XVCGPU_K_CU_STATE_H_
#define XVCGPU_K_CU_STATE_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

#define XVC_BITS_FN __device__ __forceinline__
#include "../../include/xvc_inter_bits.h"

__constant__ uint8_t kTransIdxLps[64] = {XVC_TRANS_IDX_LPS_LIST};

#define CS_R XVC_CS_MAX_REFS
#define CS_MAXCOST 0xffffffffu

__device__ __forceinline__ bool cs_affine(const xvcgpu_cs_pass &p) {
  return (p.flags & XVC_CS_AFFINE) != 0;
}
__device__ __forceinline__ bool cs_fullpel(const xvcgpu_cs_pass &p) {
  return (p.flags & XVC_CS_FULLPEL) != 0;
}
__device__ __forceinline__ bool cs_lic(const xvcgpu_cs_pass &p) {
  return (p.flags & XVC_CS_LIC) != 0;
}
// The variants of SearchMotion the folds do not run (xvcgpu_types.h): the L1 vector
// difference forced to zero (inter_search.cc:410-413, :496-518), more than one
// refinement iteration (:394).  Answered, never computed as if they were the default.
__device__ __forceinline__ bool cs_unsupported(const xvcgpu_cs_pass &p) {
  return (p.flags & XVC_CS_FORCE_L1_MVD_ZERO) != 0 || p.bi_iterations > 1;
}

// InterSearch::GetMvdBits (inter_search.cc:1149-1164): two corners for MotionVector3
__device__ __forceinline__ uint32_t cs_mvd_bits(const int32_t mvp[3][2], const int32_t mv[3][2],
                                                int down, bool affine) {
  uint32_t b = d_eg_bits((mv[0][0] - mvp[0][0]) >> (2 + down)) +
               d_eg_bits((mv[0][1] - mvp[0][1]) >> (2 + down));
  if (affine)
    b += d_eg_bits((mv[1][0] - mvp[1][0]) >> (2 + down)) +
         d_eg_bits((mv[1][1] - mvp[1][1]) >> (2 + down));
  return b;
}

// InterSearch::EvalFinalMvpIdx (:999-1020)
__device__ __forceinline__ int cs_final_mvp_idx(const int32_t mvp[2][3][2], const int32_t mv[3][2],
                                                int start, bool fullpel, bool affine) {
  const int down = fullpel ? 2 : 0;
  int best = 0;
  uint32_t best_cost = CS_MAXCOST;
  for (int i = 0; i < 2; i++) {
    const uint32_t cost = 1u + cs_mvd_bits(mvp[i], mv, down, affine);   // GetMvpBits(i, 2) = 1
    if (cost < best_cost || (cost == best_cost && i == start)) {
      best_cost = cost;
      best = i;
    }
  }
  return best;
}

// InterSearch::SetMvd (:1022-1048): MvDelta = (mv - mvp) in quarter samples
// (cu_types.h:192-194), whole-sample vectors shifted down again
__device__ __forceinline__ void cs_set_mvd(int32_t out[2][2], const int32_t mvp[3][2],
                                           const int32_t mv[3][2], bool fullpel, bool affine) {
  for (int k = 0; k < 2; k++)
    for (int c = 0; c < 2; c++) {
      int d = (k == 0 || affine) ? ((mv[k][c] - mvp[k][c]) >> 2) : 0;
      if (fullpel) d >>= 2;
      out[k][c] = d;
    }
}

__device__ __forceinline__ uint32_t cs_price(const xvcgpu_cs_pass &p, const xvc_inter_syntax &syn,
                                             uint32_t dist, uint32_t *bits_out,
                                             const xvc_bits_tables &t) {
  const uint32_t bits = xvc_inter_pred_bits(&p.ictx, &syn, &t);
  *bits_out = bits;
  return dist + ((bits * p.lambda16) >> 16);   // Bits and lambda are uint32_t (:563)
}

__device__ __forceinline__ void cs_copy_mv(int32_t dst[3][2], const int32_t src[3][2]) {
  for (int k = 0; k < 3; k++) {
    dst[k][0] = src[k][0];
    dst[k][1] = src[k][1];
  }
}

// one wave: n_words dwords global <-> LDS, all loads of a lane issued before its stores
__device__ __forceinline__ void cs_wave_copy(void *dst, const void *src, int n_words) {
  const uint32_t *s = static_cast<const uint32_t *>(src);
  uint32_t *d = static_cast<uint32_t *>(dst);
  const int lane = threadIdx.x;
  uint32_t v[2];
#pragma unroll
  for (int k = 0; k < 2; k++) v[k] = s[min(lane + 64 * k, n_words - 1)];
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (lane + 64 * k < n_words) d[lane + 64 * k] = v[k];
}
static_assert(sizeof(xvcgpu_cs_pass) % 4 == 0 && sizeof(xvcgpu_cs_pass) <= 512, "two rounds of 64 lanes");
static_assert(sizeof(xvcgpu_cs_result) % 4 == 0 && sizeof(xvcgpu_cs_result) <= 512, "two rounds of 64 lanes");

// grid: n passes; block: one wave.  `first`: the fold works on passes [first, first + n)
// of the arrays (an affine pass names its plain pass by absolute index).
#define CS_FOLD_PROLOGUE                                                         \
  __shared__ xvcgpu_cs_pass s_pass;                                              \
  __shared__ xvcgpu_cs_result s_res;                                             \
  if ((int)blockIdx.x >= n) return;                                              \
  const int pi = first + blockIdx.x;                                             \
  cs_wave_copy(&s_pass, &passes[pi], sizeof(xvcgpu_cs_pass) / 4);                \
  cs_wave_copy(&s_res, &results[pi], sizeof(xvcgpu_cs_result) / 4);              \
  __syncthreads();                                                               \
  const xvcgpu_cs_pass &p = s_pass;                                              \
  xvcgpu_cs_result &R = s_res;
// The two tables of the bit prices in LDS: a candidate's price is a chain of some twenty
// dependent look-ups by one lane (context state -> bits, -> next state), each a trip to
// the vector cache out of constant memory - most of a fold's 10 us.
#define CS_FOLD_TABLES                                                                   \
  __shared__ uint32_t s_bits[128];                                                       \
  __shared__ __attribute__((aligned(4))) uint8_t s_lps[64];                              \
  s_bits[threadIdx.x] = kEntropyBits[threadIdx.x];                                       \
  s_bits[threadIdx.x + 64] = kEntropyBits[threadIdx.x + 64];                             \
  if (threadIdx.x < 16)                                                                  \
    reinterpret_cast<uint32_t *>(s_lps)[threadIdx.x] =                                   \
        reinterpret_cast<const uint32_t *>(kTransIdxLps)[threadIdx.x];                   \
  const xvc_bits_tables tabs = {s_bits, s_lps};
#define CS_FOLD_EPILOGUE \
  __syncthreads();       \
  cs_wave_copy(&results[pi], &s_res, sizeof(xvcgpu_cs_result) / 4);

// ---- fold 1: EvalStartMvp's choice -> the searches' start predictors ---------------
// start_dist: SampleMetric(kSad) of the two predictors' predictions (GetMvpMetricType,
// :1078-1080), as xvcgpu_mc_metric_batch / xvcgpu_metric_batch return them.  Both
// candidates pay the same GetMvpBits, so the first strictly smaller distortion wins.
__device__ __forceinline__ void cs_start_fold_body(const xvcgpu_cs_pass *passes, int first, int n,
                                     const uint64_t *start_dist, xvcgpu_me_block *me_jobs,
                                     const xvcgpu_me_result *me_res,
                                     xvcgpu_affine_me_block *aff_jobs, xvcgpu_cs_result *results,
                                     int pic_w, int pic_h) {
  CS_FOLD_PROLOGUE
  auto body = [&]() {
  const bool affine = cs_affine(p);
  for (int l = 0; l < 2; l++)
    for (int r = 0; r < p.num_refs[l] && r < CS_R; r++) {
      const int sd = p.start_dist[l][r];
      const int start = sd >= 0 && start_dist[sd + 1] < start_dist[sd] ? 1 : 0;
      R.start_idx[l][r] = (uint8_t)start;
      const int j = p.uni_job[l][r];
      if (j < 0) continue;
      if (!affine) {
        xvcgpu_me_block &b = me_jobs[j];
        b.mvp_x = p.mvp[l][r][start][0][0];
        b.mvp_y = p.mvp[l][r][start][0][1];
        const int pj = p.prev_job[l][r];
        if (pj >= 0) {                       // previous_fullpel_[list][ref_idx] (:640-641)
          b.prev_x = me_res[pj].fullpel_x;
          b.prev_y = me_res[pj].fullpel_y;
        }
      } else {
        xvcgpu_affine_me_block &b = aff_jobs[j];
        cs_copy_mv(b.mvp, p.mvp[l][r][start]);
        // bootstrap: DeriveMvAffine(cu, ref, mv_normal, mv_normal) (:523-528,
        // inter_prediction.cc:615-630) = the plain pass's vector, clipped, at all corners
        int mx = results[p.plain_pass].mv[l][r][0][0], my = results[p.plain_pass].mv[l][r][0][1];
        d_clip_mv(p.x, p.y, pic_w, pic_h, mx, my);
        for (int k = 0; k < 3; k++) {
          b.bootstrap[k][0] = mx;
          b.bootstrap[k][1] = my;
        }
      }
    }
  };
  if (threadIdx.x == 0) body();
  CS_FOLD_EPILOGUE
}

__global__ void
cs_start_fold_kernel(const xvcgpu_cs_pass *passes, int first, int n,
                                     const uint64_t *start_dist, xvcgpu_me_block *me_jobs,
                                     const xvcgpu_me_result *me_res,
                                     xvcgpu_affine_me_block *aff_jobs, xvcgpu_cs_result *results,
                                     int pic_w, int pic_h) {
  cs_start_fold_body(passes, first, n, start_dist, me_jobs, me_res, aff_jobs, results, pic_w, pic_h);
}

// ---- fold 2: SearchRefIdx over both lists -> the refinement jobs --------------------
__device__ __forceinline__ void cs_uni_fold_body(const xvcgpu_cs_pass *passes, int first, int n,
                                   const xvcgpu_me_result *me_res,
                                   const xvcgpu_affine_me_result *aff_res,
                                   xvcgpu_cs_result *results, xvcgpu_bi_block *bi_jobs,
                                   xvcgpu_affine_me_block *aff_jobs) {
  CS_FOLD_TABLES
  CS_FOLD_PROLOGUE
  // the searches' results: one lane per (list, picture), the reads in flight together
  __shared__ xvcgpu_me_result s_me[2 * CS_R];
  __shared__ xvcgpu_affine_me_result s_aff[2 * CS_R];
  if (threadIdx.x < 2 * CS_R) {
    const int l = threadIdx.x / CS_R, r = threadIdx.x % CS_R;
    const int j = r < p.num_refs[l] ? p.uni_job[l][r] : -1;
    if (j >= 0) {
      if (!cs_affine(p)) s_me[threadIdx.x] = me_res[j];
      else s_aff[threadIdx.x] = aff_res[j];
    }
  }
  __syncthreads();
  // SearchRefIdx's candidates, one lane per (list, picture): final predictor, syntax,
  // price (a candidate is ~800 dependent instructions of one lane: side by side they
  // cost one candidate's time)
  if (threadIdx.x < 2 * CS_R) {
    const bool affine = cs_affine(p), fullpel = cs_fullpel(p);
    const int l = threadIdx.x / CS_R, r = threadIdx.x % CS_R;
    if (r < p.num_refs[l]) {
      const int m = l == 1 ? p.same_poc_in_l0[r] : -1;
      // list 0 searched this picture already (:536-542): its result
      const int src = m >= 0 ? m : l * CS_R + r;
      if (!affine) {
        const xvcgpu_me_result &sr = s_me[src];
        R.mv[l][r][0][0] = sr.mv_x;
        R.mv[l][r][0][1] = sr.mv_y;
        for (int k = 1; k < 3; k++) R.mv[l][r][k][0] = R.mv[l][r][k][1] = 0;
        R.dist[l][r] = sr.subpel_dist;
      } else {
        const xvcgpu_affine_me_result &sr = s_aff[src];
        cs_copy_mv(R.mv[l][r], sr.mv);
        R.dist[l][r] = sr.dist;
      }
      const int idx = cs_final_mvp_idx(p.mvp[l][r], R.mv[l][r], R.start_idx[l][r], fullpel, affine);
      R.mvp_idx[l][r] = (uint8_t)idx;
      xvc_inter_syntax syn = {};
      syn.inter_dir = (uint8_t)l;
      syn.use_affine = affine;
      syn.fullpel_mv = fullpel;
      syn.use_lic = cs_lic(p);
      syn.ref_idx[l] = (int8_t)r;
      syn.mvp_idx[l] = (uint8_t)idx;
      cs_set_mvd(syn.mvd[l], p.mvp[l][r][idx], R.mv[l][r], fullpel, affine);
      R.cost[l][r] = cs_price(p, syn, R.dist[l][r], &R.bits[l][r], tabs);
    }
  }
  __syncthreads();
  auto body = [&]() {
  const bool affine = cs_affine(p), fullpel = cs_fullpel(p);
  R.cost_l1_unique = CS_MAXCOST;
  R.best_ref_l1_unique = -1;
  for (int l = 0; l < 2; l++) {          // the folds over the candidates, in the reference's order
    uint32_t cost_best = CS_MAXCOST;
    int best = -1;
    for (int r = 0; r < p.num_refs[l] && r < CS_R; r++) {
      const int m = l == 1 ? p.same_poc_in_l0[r] : -1;
      const uint32_t cost = R.cost[l][r];
      if (cost < cost_best) {
        cost_best = cost;
        best = r;
      }
      if (l == 1 && m < 0 && cost < R.cost_l1_unique) {
        R.cost_l1_unique = cost;
        R.best_ref_l1_unique = (int8_t)r;
      }
    }
    R.cost_list[l] = cost_best;
    R.best_ref[l] = (int8_t)best;
  }
  // the refinement job slots of the pass: all empty, then the ones this state runs
  const int slots = 2 * CS_R * CS_R;
  for (int i = 0; i < slots; i++) {
    if (!affine)
      bi_jobs[p.bi_job + i].blk.w = 0;     // not a block size: the kernels skip it
    else
      aff_jobs[p.bi_job + i].w = 0;
  }
  R.bi_valid = 0;
  if (cs_unsupported(p)) {                 // no refinement job: the bi fold answers
    R.which = XVC_CS_WHICH_UNSUPPORTED;
    return;
  }
  if (!p.num_refs[1]) return;              // kUniPredOnly (:228-230)
  // SearchBiIterative (:392-433), one iteration: searches the list that lost
  const int best_dir = R.cost_list[0] <= R.cost_list[1] ? 0 : 1;
  const int s = 1 - best_dir, o = R.best_ref[best_dir];
  R.search_list = (uint8_t)s;
  R.bi_valid = 1;
  for (int r = 0; r < p.num_refs[s] && r < CS_R; r++) {
    const int slot = p.bi_job + (s * CS_R + r) * CS_R + o;
    const int idx = R.mvp_idx[s][r];       // unipred_best_mvp_idx_ (:497-499)
    if (!affine) {
      xvcgpu_bi_block &j = bi_jobs[slot];
      j.blk.x = p.x;
      j.blk.y = p.y;
      j.blk.w = p.w;
      j.blk.h = p.h;
      j.blk.depth_nonzero = 0;
      j.blk.fullpel_mv = fullpel ? XVC_ME_FULLPEL_MV : 0;   // (a LIC pass: xvcgpu_bipred_search_lic)
      j.blk.mvp_x = p.mvp[s][r][idx][0][0];
      j.blk.mvp_y = p.mvp[s][r][idx][0][1];
      j.blk.prev_x = j.blk.prev_y = 0;
      j.blk.lambda16 = p.lambda16;
      j.blk.search_range = 4;              // inter_search_range_bi
      j.other_mv_x = R.mv[best_dir][o][0][0];
      j.other_mv_y = R.mv[best_dir][o][0][1];
      j.boot_mv_x = R.mv[s][r][0][0];      // GetBestUniPredMv (:499)
      j.boot_mv_y = R.mv[s][r][0][1];
    } else {
      xvcgpu_affine_me_block &j = aff_jobs[slot];
      j.x = p.x;
      j.y = p.y;
      j.w = p.w;
      j.h = p.h;
      j.flags = XVC_AFFINE_ME_HAS_BOOTSTRAP | XVC_AFFINE_ME_BIPRED;
      j.reserved = 0;
      j.lambda16 = p.lambda16;
      cs_copy_mv(j.mvp, p.mvp[s][r][idx]);
      cs_copy_mv(j.bootstrap, R.mv[s][r]);
      cs_copy_mv(j.other_mv, R.mv[best_dir][o]);
    }
  }
  };
  if (threadIdx.x == 0) body();
  CS_FOLD_EPILOGUE
}

__global__ void
cs_uni_fold_kernel(const xvcgpu_cs_pass *passes, int first, int n,
                                   const xvcgpu_me_result *me_res,
                                   const xvcgpu_affine_me_result *aff_res,
                                   xvcgpu_cs_result *results, xvcgpu_bi_block *bi_jobs,
                                   xvcgpu_affine_me_block *aff_jobs) {
  cs_uni_fold_body(passes, first, n, me_res, aff_res, results, bi_jobs, aff_jobs);
}

// ---- fold 3: the refinement's costs, the three-way choice, the evaluation's jobs ----
__device__ __forceinline__ void cs_bi_fold_body(const xvcgpu_cs_pass *passes, int first, int n,
                                  const xvcgpu_me_result *bi_res,
                                  const xvcgpu_affine_me_result *aff_res,
                                  xvcgpu_cs_result *results, xvcgpu_inter_block *ev_inter) {
  CS_FOLD_TABLES
  CS_FOLD_PROLOGUE
  __shared__ xvcgpu_me_result s_bi[CS_R];
  __shared__ xvcgpu_affine_me_result s_abi[CS_R];
  if (threadIdx.x < CS_R && R.bi_valid) {
    const int s = R.search_list, r = threadIdx.x;
    if (r < p.num_refs[s]) {
      const int slot = p.bi_job + (s * CS_R + r) * CS_R + R.best_ref[1 - s];
      if (!cs_affine(p)) s_bi[r] = bi_res[slot];
      else s_abi[r] = aff_res[slot];
    }
  }
  __syncthreads();
  // the refinement's candidates, one lane per searched picture
  if (threadIdx.x < CS_R && R.bi_valid) {
    const bool affine = cs_affine(p), fullpel = cs_fullpel(p);
    const int s = R.search_list, od = 1 - s, o = R.best_ref[od], r = threadIdx.x;
    if (r < p.num_refs[s]) {
      int32_t other_mvd[2][2];
      cs_set_mvd(other_mvd, p.mvp[od][o][R.mvp_idx[od][o]], R.mv[od][o], fullpel, affine);
      if (!affine) {
        R.bi_mv[r][0][0] = s_bi[r].mv_x;
        R.bi_mv[r][0][1] = s_bi[r].mv_y;
        for (int k = 1; k < 3; k++) R.bi_mv[r][k][0] = R.bi_mv[r][k][1] = 0;
        R.bi_dist[r] = s_bi[r].subpel_dist;
      } else {
        cs_copy_mv(R.bi_mv[r], s_abi[r].mv);
        R.bi_dist[r] = s_abi[r].dist;
      }
      const int idx = cs_final_mvp_idx(p.mvp[s][r], R.bi_mv[r], R.mvp_idx[s][r], fullpel, affine);
      R.bi_mvp_idx[r] = (uint8_t)idx;
      xvc_inter_syntax syn = {};
      syn.inter_dir = 2;
      syn.use_affine = affine;
      syn.fullpel_mv = fullpel;
      syn.use_lic = cs_lic(p);
      syn.ref_idx[s] = (int8_t)r;
      syn.mvp_idx[s] = (uint8_t)idx;
      cs_set_mvd(syn.mvd[s], p.mvp[s][r][idx], R.bi_mv[r], fullpel, affine);
      syn.ref_idx[od] = (int8_t)o;
      syn.mvp_idx[od] = R.mvp_idx[od][o];
      for (int k = 0; k < 2; k++)
        for (int c = 0; c < 2; c++) syn.mvd[od][k][c] = other_mvd[k][c];
      R.bi_cost[r] = cs_price(p, syn, R.bi_dist[r], &R.bi_bits[r], tabs);
    }
  }
  __syncthreads();
  auto body = [&]() {
  const bool affine = cs_affine(p), fullpel = cs_fullpel(p);
  if (cs_unsupported(p)) {
    // answered, not computed: no motion, never chosen, the evaluation gets no reference
    R.which = XVC_CS_WHICH_UNSUPPORTED;
    R.chosen = 0;
    R.best_cost = CS_MAXCOST;
    R.inter_dir = 0;
    R.ref_idx[0] = R.ref_idx[1] = -1;
    if (p.eval >= 0)
      for (int c = 0; c < 3; c++) {
        xvcgpu_inter_block &b = ev_inter[3 * p.eval + c];
        b.ref[0] = b.ref[1] = -1;
      }
    return;
  }
  uint32_t cost_bi = CS_MAXCOST;
  int bi_ref = -1;
  const int s = R.search_list, od = 1 - s;
  const int o = R.bi_valid ? R.best_ref[od] : -1;
  if (R.bi_valid)
    for (int r = 0; r < p.num_refs[s] && r < CS_R; r++)
      if (R.bi_cost[r] < cost_bi) {
        cost_bi = R.bi_cost[r];
        bi_ref = r;
      }
  // the three-way choice (:247-257); a picture with one list returns list 0's result
  const uint32_t c0 = R.cost_list[0], c1u = R.cost_l1_unique;
  int which;
  if (!p.num_refs[1])
    which = 1;
  else if (cost_bi <= c0 && cost_bi <= c1u)
    which = 0;
  else if (c0 <= c1u)
    which = 1;
  else
    which = 2;
  R.which = (uint8_t)which;
  for (int l = 0; l < 2; l++) {
    R.ref_idx[l] = -1;
    R.out_mvp_idx[l] = 0;
    for (int k = 0; k < 3; k++) R.out_mv[l][k][0] = R.out_mv[l][k][1] = 0;
    for (int k = 0; k < 2; k++) R.out_mvd[l][k][0] = R.out_mvd[l][k][1] = 0;
  }
  auto take_uni = [&](int l, int r) {
    R.ref_idx[l] = (int8_t)r;
    R.out_mvp_idx[l] = R.mvp_idx[l][r];
    cs_copy_mv(R.out_mv[l], R.mv[l][r]);
    cs_set_mvd(R.out_mvd[l], p.mvp[l][r][R.mvp_idx[l][r]], R.mv[l][r], fullpel, affine);
  };
  if (which == 0) {
    R.inter_dir = 2;
    R.best_cost = cost_bi;
    take_uni(od, o);
    R.ref_idx[s] = (int8_t)bi_ref;
    R.out_mvp_idx[s] = R.bi_mvp_idx[bi_ref];
    cs_copy_mv(R.out_mv[s], R.bi_mv[bi_ref]);
    cs_set_mvd(R.out_mvd[s], p.mvp[s][bi_ref][R.bi_mvp_idx[bi_ref]], R.bi_mv[bi_ref], fullpel, affine);
  } else if (which == 1) {
    R.inter_dir = 0;
    R.best_cost = c0;
    take_uni(0, R.best_ref[0]);
  } else {
    R.inter_dir = 1;
    R.best_cost = c1u;
    take_uni(1, R.best_ref_l1_unique);
  }
  // CodingUnit::HasZeroMvd (coding_unit.cc:445-453)
  if (R.inter_dir == 2)
    R.zero_mvd = !(R.out_mvd[0][0][0] | R.out_mvd[0][0][1] | R.out_mvd[1][0][0] | R.out_mvd[1][0][1]);
  else
    R.zero_mvd = !(R.out_mvd[R.inter_dir][0][0] | R.out_mvd[R.inter_dir][0][1]);
  // CompressInter (:80-93): the affine pass replaces the plain result only when cheaper
  R.chosen = 1;
  const xvcgpu_cs_result *final = &R;
  const xvcgpu_cs_pass *fp = &p;
  if (affine && p.plain_pass >= 0) {
    if (results[p.plain_pass].best_cost <= R.best_cost) {
      R.chosen = 0;
      final = &results[p.plain_pass];
      fp = &passes[p.plain_pass];
    } else {
      results[p.plain_pass].chosen = 0;
    }
  }
  if (p.eval < 0) return;
  const bool fa = cs_affine(*fp);
  for (int c = 0; c < 3; c++) {
    xvcgpu_inter_block &b = ev_inter[3 * p.eval + c];
    b.flags = fa ? XVC_INTER_AFFINE : (cs_lic(*fp) ? XVC_INTER_LIC : 0);
    for (int l = 0; l < 2; l++) {
      const bool used = final->inter_dir == 2 || final->inter_dir == l;
      b.ref[l] = used ? fp->slot[l][final->ref_idx[l]] : -1;
      for (int k = 0; k < 3; k++) {
        b.mv[l][k][0] = used ? final->out_mv[l][k][0] : 0;
        b.mv[l][k][1] = used ? final->out_mv[l][k][1] : 0;
      }
    }
  }
  };
  if (threadIdx.x == 0) body();
  CS_FOLD_EPILOGUE
}

__global__ void
cs_bi_fold_kernel(const xvcgpu_cs_pass *passes, int first, int n,
                                  const xvcgpu_me_result *bi_res,
                                  const xvcgpu_affine_me_result *aff_res,
                                  xvcgpu_cs_result *results, xvcgpu_inter_block *ev_inter) {
  cs_bi_fold_body(passes, first, n, bi_res, aff_res, results, ev_inter);
}

// ---- the merge ranking's fold: SearchMergeCandidates' arithmetic (:176-196) ---------
// One thread per ranking: five costs in double (dist + bits * lambda_sqrt as the
// reference forms it - no contraction, -ffp-contract=off), std::stable_sort = an
// insertion sort that only moves on strictly smaller, the 1.25 x cut from the back.
__device__ __forceinline__ void cs_merge_fold_body(const xvcgpu_cs_merge *merges, int first, int n,
                                     const uint64_t *dist, const xvcgpu_inter_block *cands,
                                     xvcgpu_cs_merge_result *results,
                                     xvcgpu_inter_block *ev_inter) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const xvcgpu_cs_merge m = merges[first + i];
  double cost[XVC_CS_MERGE_CANDS];
  int order[XVC_CS_MERGE_CANDS];
  for (int k = 0; k < XVC_CS_MERGE_CANDS; k++) {
    const int bits = k + 1 - (k < XVC_CS_MERGE_CANDS - 1 ? 0 : 1);
    cost[k] = (double)dist[m.dist + k] + (double)bits * m.lambda_sqrt;
    order[k] = k;
  }
  for (int a = 1; a < XVC_CS_MERGE_CANDS; a++) {   // stable: equal costs keep their order
    const double c = cost[a];
    const int o = order[a];
    int b = a - 1;
    while (b >= 0 && c < cost[b]) {
      cost[b + 1] = cost[b];
      order[b + 1] = order[b];
      b--;
    }
    cost[b + 1] = c;
    order[b + 1] = o;
  }
  int num = XVC_CS_MERGE_SLOTS;
  for (int k = XVC_CS_MERGE_SLOTS; k >= 0; k--)
    if (cost[k] > cost[0] * 1.25) num = k;
  xvcgpu_cs_merge_result r;
  for (int k = 0; k < XVC_CS_MERGE_CANDS; k++) {
    r.cost[k] = cost[k];
    r.order[k] = order[k];
  }
  r.num = num;
  r.reserved[0] = r.reserved[1] = 0;
  results[first + i] = r;
  if (m.slot < 0) return;
  for (int sl = 0; sl < XVC_CS_MERGE_SLOTS; sl++) {
    const xvcgpu_inter_block src = cands[m.cand + order[sl]];
    for (int c = 0; c < 3; c++) {
      xvcgpu_inter_block &b = ev_inter[3 * (m.slot + sl) + c];
      const bool on = sl < num;
      b.flags = on ? src.flags : 0;
      for (int l = 0; l < 2; l++) {
        b.ref[l] = on ? src.ref[l] : (int8_t)-1;
        for (int k = 0; k < 3; k++) {
          b.mv[l][k][0] = on ? src.mv[l][k][0] : 0;
          b.mv[l][k][1] = on ? src.mv[l][k][1] : 0;
        }
      }
    }
  }
}

__global__ void
cs_merge_fold_kernel(const xvcgpu_cs_merge *merges, int first, int n,
                                     const uint64_t *dist, const xvcgpu_inter_block *cands,
                                     xvcgpu_cs_merge_result *results,
                                     xvcgpu_inter_block *ev_inter) {
  cs_merge_fold_body(merges, first, n, dist, cands, results, ev_inter);
}

// ---- the distortions of an evaluation in one launch ----------------------------------
// metric_batch_kernel (k_metric.h) with the planes chosen per candidate: an evaluation
// compares three components against two pictures with three weights - seven launches
// of 5 us each through xvcgpu_metric_batch, a fifth of a CU state's time.
// grid: ceil(n/4); block: 256 = 4 waves, one candidate per wave.
__device__ __forceinline__ void eval_dist_body(PicView orig, PicView pred, PicView rec, int strength,
                 const xvcgpu_eval_cand *cands, int n, uint64_t *out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n) return;
  const xvcgpu_eval_cand cd = cands[c];
  const PlaneView pa = orig.c[cd.comp], pb = cd.versus ? rec.c[cd.comp] : pred.c[cd.comp];
  const uint16_t *a = cd.orig_at ? pa.p + (ptrdiff_t)cd.oy * pa.stride + cd.ox
                                 : pa.p + (ptrdiff_t)cd.y * pa.stride + cd.x;
  const uint16_t *b = pb.p + (ptrdiff_t)cd.y * pb.stride + cd.x;
  const uint64_t dist = wave_compare(cd.metric, orig.bd, cd.qp, strength, cd.w, cd.h, a,
                                     pa.stride, b, pb.stride);
  if ((threadIdx.x & 63) == 0) out[c] = (uint64_t)((double)dist * cd.weight);
}

__global__ void __launch_bounds__(256)
eval_dist_kernel(PicView orig, PicView pred, PicView rec, int strength,
                 const xvcgpu_eval_cand *cands, int n, uint64_t *out) {
  eval_dist_body(orig, pred, rec, strength, cands, n, out);
}

#endif  // XVCGPU_K_CU_STATE_H_
