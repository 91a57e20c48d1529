// xvc_inter_search.cc -- C entry points (for ctypes / tests) of the host control
// that xvc_gpu::InterSearch (xvc_gpu_ops.h) keeps around the motion-search
// batches: EvalStartMvp, EvalFinalMvpIdx, the per-list SearchRefIdx loop, the merge
// fold, SearchBiIterative and SearchMotion (inter_search.cc:165-259, :392-578,
// :966-1019).  Handles are borrowed, never freed here.
#include <array>
#include <cstdint>
#include <memory>
#include <vector>

#include "xvc_gpu_ops.h"

namespace {
typedef std::array<int32_t, 4> Mvp;
std::vector<Mvp> MvpList(const int32_t *mvp, int n) {
  std::vector<Mvp> out(n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 4; k++) out[i][k] = mvp[4 * i + k];
  return out;
}
}  // namespace

extern "C" {

// blocks: host array of n xvcgpu_me_block; mvp: n x {x0, y0, x1, y1};
// out_idx / out_cost: n each.
int xvc_host_eval_start_mvp_batch(xvcgpu_ctx *ctx, xvcgpu_picture *orig, xvcgpu_picture *ref,
                                  const xvcgpu_me_block *blocks, int n, const int32_t *mvp,
                                  int32_t *out_idx, uint32_t *out_cost) {
  if (!ctx || !orig || !ref || !blocks || !mvp || n < 0) return XVCGPU_INVALID_ARGUMENT;
  try {
    xvc_gpu::Context c(ctx);
    xvc_gpu::Picture o(c, orig), r(c, ref);
    const std::vector<xvcgpu_me_block> b(blocks, blocks + n);
    const std::vector<xvc_gpu::InterSearch::StartMvp> res =
        xvc_gpu::InterSearch(c).EvalStartMvpBatch(o, r, b, MvpList(mvp, n));
    for (int i = 0; i < n; i++) {
      if (out_idx) out_idx[i] = res[i].idx;
      if (out_cost) out_cost[i] = res[i].cost;
    }
    return XVCGPU_OK;
  } catch (const xvc_gpu::Error &e) {
    return e.status;
  }
}

int xvc_host_eval_final_mvp_idx(const int32_t mvp[4], int mv_x, int mv_y, int start,
                                int fullpel_mv) {
  Mvp m = {{mvp[0], mvp[1], mvp[2], mvp[3]}};
  return xvc_gpu::InterSearch::EvalFinalMvpIdx(m, mv_x, mv_y, start, fullpel_mv != 0);
}

uint32_t xvc_host_mvd_bits(int mvp_x, int mvp_y, int mv_x, int mv_y, int down_shift) {
  return xvc_gpu::InterSearch::GetMvdBits(mvp_x, mvp_y, mv_x, mv_y, down_shift);
}

// The per-list loop over n_refs reference pictures: mvp[r * n + i][4],
// side_bits[r * n + i]; out: per CU {ref_idx, mvp_idx, mv_x, mv_y, dist, cost}.
int xvc_host_search_ref_idx_batch(xvcgpu_ctx *ctx, xvcgpu_picture *orig,
                                  xvcgpu_picture *const *refs, int n_refs,
                                  const xvcgpu_me_block *blocks, int n, const int32_t *mvp,
                                  const uint32_t *side_bits, int32_t *out) {
  if (!ctx || !orig || !refs || !blocks || !mvp || !side_bits || !out || n < 0 || n_refs < 1)
    return XVCGPU_INVALID_ARGUMENT;
  try {
    xvc_gpu::Context c(ctx);
    xvc_gpu::Picture o(c, orig);
    std::vector<std::unique_ptr<xvc_gpu::Picture>> views;
    std::vector<const xvc_gpu::Picture *> ref_pics;
    std::vector<std::vector<Mvp>> lists;
    std::vector<std::vector<uint32_t>> bits;
    for (int r = 0; r < n_refs; r++) {
      views.emplace_back(new xvc_gpu::Picture(c, refs[r]));
      ref_pics.push_back(views.back().get());
      lists.push_back(MvpList(mvp + 4 * static_cast<size_t>(r) * n, n));
      bits.push_back(std::vector<uint32_t>(side_bits + static_cast<size_t>(r) * n,
                                           side_bits + static_cast<size_t>(r + 1) * n));
    }
    const std::vector<xvcgpu_me_block> b(blocks, blocks + n);
    const std::vector<xvc_gpu::InterSearch::UniPredChoice> res =
        xvc_gpu::InterSearch(c).SearchRefIdxBatch(o, ref_pics, b, lists, bits);
    for (int i = 0; i < n; i++) {
      int32_t *q = out + 6 * i;
      q[0] = res[i].ref_idx;
      q[1] = res[i].mvp_idx;
      q[2] = res[i].mv_x;
      q[3] = res[i].mv_y;
      q[4] = static_cast<int32_t>(res[i].dist);
      q[5] = static_cast<int32_t>(res[i].cost);
    }
    return XVCGPU_OK;
  } catch (const xvc_gpu::Error &e) {
    return e.status;
  }
}

// SearchMergeCandidates for n CUs: cands[n][5][5] = {inter_dir, mv0_x, mv0_y,
// mv1_x, mv1_y}; out[n][6] = the five candidate indices cheapest first, then the
// number to try.  `pred`: a scratch picture.
int xvc_host_search_merge_candidates_batch(xvcgpu_ctx *ctx, xvcgpu_picture *orig,
                                           xvcgpu_picture *ref_l0, xvcgpu_picture *ref_l1,
                                           xvcgpu_picture *pred, const xvcgpu_me_block *blocks,
                                           int n, const int32_t *cands,
                                           const double *lambda_sqrt, int32_t *out) {
  if (!ctx || !orig || !ref_l0 || !ref_l1 || !pred || !blocks || !cands || !lambda_sqrt || !out ||
      n < 0)
    return XVCGPU_INVALID_ARGUMENT;
  try {
    xvc_gpu::Context c(ctx);
    xvc_gpu::Picture o(c, orig), r0(c, ref_l0), r1(c, ref_l1), p(c, pred);
    constexpr int K = xvc_gpu::InterSearch::kNumMergeCand;
    std::vector<std::array<xvc_gpu::InterSearch::MergeCand, K>> list(n);
    for (int i = 0; i < n; i++)
      for (int m = 0; m < K; m++) {
        const int32_t *q = cands + 5 * K * static_cast<size_t>(i) + 5 * m;
        xvc_gpu::InterSearch::MergeCand mc = {q[0], {{q[1], q[2]}, {q[3], q[4]}}};
        list[i][m] = mc;
      }
    const std::vector<xvcgpu_me_block> b(blocks, blocks + n);
    const std::vector<double> ls(lambda_sqrt, lambda_sqrt + n);
    const std::vector<xvc_gpu::InterSearch::MergeChoice> res =
        xvc_gpu::InterSearch(c).SearchMergeCandidatesBatch(o, r0, r1, &p, b, list, ls);
    for (int i = 0; i < n; i++) {
      for (int m = 0; m < K; m++) out[(K + 1) * i + m] = res[i].order[m];
      out[(K + 1) * i + K] = res[i].num;
    }
    return XVCGPU_OK;
  } catch (const xvc_gpu::Error &e) {
    return e.status;
  }
}

// InterSearch::SearchMotion for n CUs of a bi-predicted picture, one reference per
// list: blocks[l * n + i] (the descriptor of CU i for list l: search range and
// previous vector are per list), mvp[(l * n + i) * 4], side_bits_uni[l * n + i],
// side_bits_bi[i]; out[n][18] = {inter_dir, mv0_x, mv0_y, mv1_x, mv1_y, mvp_idx0,
// mvp_idx1, cost; then the parts: {cost, mv_x, mv_y, mvp_idx} of the L0 and of the
// L1 uni-directional search, the bi-directional cost, the refinement steps taken}.
int xvc_host_search_motion_batch(xvcgpu_ctx *ctx, xvcgpu_picture *orig, xvcgpu_picture *ref_l0,
                                 xvcgpu_picture *ref_l1, const xvcgpu_me_block *blocks, int n,
                                 const int32_t *mvp, const uint32_t *side_bits_uni,
                                 const uint32_t *side_bits_bi, int num_iterations,
                                 int64_t *out) {
  if (!ctx || !orig || !ref_l0 || !ref_l1 || !blocks || !mvp || !side_bits_uni ||
      !side_bits_bi || !out || n < 0 || num_iterations < 1)
    return XVCGPU_INVALID_ARGUMENT;
  try {
    xvc_gpu::Context c(ctx);
    xvc_gpu::Picture o(c, orig), r0(c, ref_l0), r1(c, ref_l1);
    std::vector<xvcgpu_me_block> b[2];
    std::vector<Mvp> m[2];
    std::vector<uint32_t> su[2];
    for (int l = 0; l < 2; l++) {
      b[l].assign(blocks + static_cast<size_t>(l) * n, blocks + static_cast<size_t>(l + 1) * n);
      m[l] = MvpList(mvp + 4 * static_cast<size_t>(l) * n, n);
      su[l].assign(side_bits_uni + static_cast<size_t>(l) * n,
                   side_bits_uni + static_cast<size_t>(l + 1) * n);
    }
    const std::vector<uint32_t> sb(side_bits_bi, side_bits_bi + n);
    const std::vector<xvc_gpu::InterSearch::MotionChoice> res =
        xvc_gpu::InterSearch(c).SearchMotionBatch(o, r0, r1, b, m, su, sb, num_iterations);
    for (int i = 0; i < n; i++) {
      int64_t *q = out + 18 * i;
      q[0] = res[i].inter_dir;
      q[1] = res[i].list[0].mv_x;
      q[2] = res[i].list[0].mv_y;
      q[3] = res[i].list[1].mv_x;
      q[4] = res[i].list[1].mv_y;
      q[5] = res[i].list[0].mvp_idx;
      q[6] = res[i].list[1].mvp_idx;
      q[7] = res[i].cost;
      for (int l = 0; l < 2; l++) {
        q[8 + 4 * l] = res[i].cost_uni[l];
        q[9 + 4 * l] = res[i].uni[l].mv_x;
        q[10 + 4 * l] = res[i].uni[l].mv_y;
        q[11 + 4 * l] = res[i].uni[l].mvp_idx;
      }
      q[16] = res[i].cost_bi;
      q[17] = res[i].bi_steps;
    }
    return XVCGPU_OK;
  } catch (const xvc_gpu::Error &e) {
    return e.status;
  }
}

// InterSearch::SearchMotion with up to three reference pictures per list
// (xvc_gpu::InterSearch::SearchMotionMultiBatch): ref_pics[l * 3 + r] (NULL beyond
// num_ref[l]), same_poc_in_l0[r] for the list-1 pictures, blocks[(l * 3 + r) * n + i],
// mvp[((l * 3 + r) * n + i) * 4]; out[n][32] = {inter_dir, cost, then per list
// {ref_idx, mvp_idx, mv_x, mv_y} at [2 + 4 l], cost_l0, cost_l1, cost_l1_unique,
// cost_bi at [10..13], the lists' uni-directional choices {ref_idx, mvp_idx, mv_x,
// mv_y} at [14 + 4 l], bi_steps at [22]}.
int xvc_host_search_motion_multi_batch(xvcgpu_ctx *ctx, xvcgpu_picture *orig,
                                       xvcgpu_picture *const *ref_pics, const int32_t *num_ref,
                                       const int32_t *same_poc_in_l0, int pic_is_uni,
                                       int force_l1_mvd_zero, const xvcgpu_me_block *blocks, int n,
                                       const int32_t *mvp, int num_iterations, int64_t *out) {
  typedef xvc_gpu::InterSearch IS;
  if (!ctx || !orig || !ref_pics || !num_ref || !same_poc_in_l0 || !blocks || !mvp || !out ||
      n < 0 || num_iterations < 1 || num_ref[0] < 1 || num_ref[0] > IS::kMaxRefs ||
      num_ref[1] < 1 || num_ref[1] > IS::kMaxRefs)
    return XVCGPU_INVALID_ARGUMENT;
  try {
    xvc_gpu::Context c(ctx);
    xvc_gpu::Picture o(c, orig);
    std::vector<std::unique_ptr<xvc_gpu::Picture>> views;
    IS::RefLists refs;
    std::vector<xvcgpu_me_block> b[2][IS::kMaxRefs];
    std::vector<Mvp> m[2][IS::kMaxRefs];
    for (int l = 0; l < 2; l++) {
      refs.num_ref[l] = num_ref[l];
      for (int r = 0; r < IS::kMaxRefs; r++) {
        refs.pic[l][r] = nullptr;
        if (r >= num_ref[l]) continue;
        if (!ref_pics[l * IS::kMaxRefs + r]) return XVCGPU_INVALID_ARGUMENT;
        views.emplace_back(new xvc_gpu::Picture(c, ref_pics[l * IS::kMaxRefs + r]));
        refs.pic[l][r] = views.back().get();
        const size_t at = static_cast<size_t>(l * IS::kMaxRefs + r) * n;
        b[l][r].assign(blocks + at, blocks + at + n);
        m[l][r] = MvpList(mvp + 4 * at, n);
      }
    }
    for (int r = 0; r < IS::kMaxRefs; r++) {
      refs.same_poc_in_l0[r] = r < num_ref[1] ? same_poc_in_l0[r] : -1;
      if (refs.same_poc_in_l0[r] >= num_ref[0]) return XVCGPU_INVALID_ARGUMENT;
    }
    refs.pic_is_uni = pic_is_uni != 0;
    refs.force_l1_mvd_zero = force_l1_mvd_zero != 0;
    const std::vector<IS::MotionChoiceMulti> res =
        IS(c).SearchMotionMultiBatch(o, refs, b, m, num_iterations);
    for (int i = 0; i < n; i++) {
      int64_t *q = out + 32 * i;
      for (int k = 0; k < 32; k++) q[k] = 0;
      q[0] = res[i].inter_dir;
      q[1] = res[i].cost;
      for (int l = 0; l < 2; l++) {
        q[2 + 4 * l] = res[i].list[l].ref_idx;
        q[3 + 4 * l] = res[i].list[l].mvp_idx;
        q[4 + 4 * l] = res[i].list[l].mv_x;
        q[5 + 4 * l] = res[i].list[l].mv_y;
        q[14 + 4 * l] = res[i].uni[l].ref_idx;
        q[15 + 4 * l] = res[i].uni[l].mvp_idx;
        q[16 + 4 * l] = res[i].uni[l].mv_x;
        q[17 + 4 * l] = res[i].uni[l].mv_y;
      }
      q[10] = res[i].cost_l0;
      q[11] = res[i].cost_l1;
      q[12] = res[i].cost_l1_unique;
      q[13] = res[i].cost_bi;
      q[22] = res[i].bi_steps;
    }
    return XVCGPU_OK;
  } catch (const xvc_gpu::Error &e) {
    return e.status;
  }
}

int xvc_host_choose_uni_or_bi(uint32_t cost_l0, uint32_t cost_l1_unique, uint32_t cost_bi) {
  return xvc_gpu::InterSearch::ChooseUniOrBi(cost_l0, cost_l1_unique, cost_bi);
}

}  // extern "C"
