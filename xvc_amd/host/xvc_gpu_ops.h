// xvc_gpu_ops.h -- header-only C++11 host layer over the C-ABI (include/xvcgpu.h),
// carrying the reference's operator names and argument meaning so that the
// reference-side binding reads like the code it replaces:
//
//   reference class / call                         here
//   ---------------------------------------------  ----------------------------
//   YuvPicture + PadBorder (yuv_pic.cc:32-150)      xvc_gpu::Picture
//   SampleMetric::Compare (sample_metric.cc:171)    xvc_gpu::SampleMetric::CompareBatch
//   InterSearch::MotionEstNormal (inter_search.cc:606)  xvc_gpu::InterSearch::MotionEstNormalBatch
//   InterSearch::SearchMotion (:198-259),           xvc_gpu::InterSearch::SearchMotionBatch,
//     SearchBiIterative (:392-433)                    SearchBiIterativeBatch
//   InterPrediction::MotionCompensationMv (:740)    xvc_gpu::InterPrediction::MotionCompensationBatch
//   TransformEncoder::TransformAndReconstruct       xvc_gpu::TransformEncoder::TransformAndReconstructBatch
//     (transform_encoder.cc:203)
//   DeblockingFilter::DeblockPicture                xvc_gpu::DeblockingFilter::DeblockPicture
//     (deblocking_filter.cc:56)
//   Resampler::ConvertFrom / ConvertTo              xvc_gpu::Resampler::ConvertFrom / ConvertTo
//     (resample.cc:32, :96; no resizing)
//   Checksum(kCrc, mode)::HashPicture               xvc_gpu::Checksum::HashPicture
//     (checksum.cc:30-92)
//   CuEncoder::CalcDeltaQpFromVariance              xvc_gpu::AdaptiveQp::CalcDeltaQpFromVariance
//     (cu_encoder.cc:308-363)
//   PictureEncoder::DetermineAllowLic               xvc_gpu::DetermineAllowLic
//     (picture_encoder.cc:230-281)
//   IntraPrediction::FillReferenceState + Predict   xvc_gpu::IntraPrediction::PredictBatch
//     (intra_prediction.cc:81-147)
//   IntraSearch::DetermineSlowIntraModes, the       xvc_gpu::IntraSearch::SatdAllModesBatch
//     prediction + SATD loop (intra_search.cc:189-305)
//
// Errors: the reference asserts internally and returns enum codes at its C API
// (xvcenc.h:34-45); here every failing xvcgpu call throws xvc_gpu::Error inside
// the host layer only (never across the C-ABI).  No CPU fallback exists.
#ifndef XVC_AMD_HOST_XVC_GPU_OPS_H_
#define XVC_AMD_HOST_XVC_GPU_OPS_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "xvcgpu.h"

namespace xvc_gpu {

struct Error : std::runtime_error {
  Error(xvcgpu_status st, const std::string &what)
      : std::runtime_error(what), status(st) {}
  xvcgpu_status status;
};

class Context {
 public:
  explicit Context(int device = 0) : ctx_(nullptr), owned_(true) {
    xvcgpu_status st = xvcgpu_create(device, &ctx_);
    if (st != XVCGPU_OK) throw Error(st, "xvcgpu_create: no gfx950 device");
  }
  // a view of a context somebody else owns
  explicit Context(xvcgpu_ctx *borrowed) : ctx_(borrowed), owned_(false) {}
  ~Context() {
    if (owned_) xvcgpu_destroy(ctx_);
  }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  xvcgpu_ctx *get() const { return ctx_; }
  void Check(xvcgpu_status st) const {
    if (st != XVCGPU_OK) throw Error(st, xvcgpu_last_error(ctx_));
  }
  void Sync() const { Check(xvcgpu_sync(ctx_)); }

 private:
  xvcgpu_ctx *ctx_;
  bool owned_;
};

// Typed device array (descriptors in, results out).
template <typename T>
class DeviceArray {
 public:
  DeviceArray(const Context &ctx, size_t n) : ctx_(ctx), n_(n), ptr_(nullptr) {
    ctx_.Check(xvcgpu_malloc(ctx_.get(), n * sizeof(T), &ptr_));
  }
  DeviceArray(const Context &ctx, const std::vector<T> &host)
      : DeviceArray(ctx, host.size()) {
    ctx_.Check(xvcgpu_memcpy_h2d(ctx_.get(), ptr_, host.data(), n_ * sizeof(T)));
  }
  ~DeviceArray() { xvcgpu_free(ctx_.get(), ptr_); }
  DeviceArray(const DeviceArray &) = delete;
  DeviceArray &operator=(const DeviceArray &) = delete;
  T *data() const { return static_cast<T *>(ptr_); }
  size_t size() const { return n_; }
  std::vector<T> ToHost() const {
    std::vector<T> out(n_);
    ctx_.Check(xvcgpu_memcpy_d2h(ctx_.get(), out.data(), ptr_, n_ * sizeof(T)));
    return out;
  }

 private:
  const Context &ctx_;
  size_t n_;
  void *ptr_;
};

// Device twin of YuvPicture.
class Picture {
 public:
  Picture(const Context &ctx, int width, int height, int bitdepth)
      : ctx_(ctx), pic_(nullptr), owned_(true) {
    ctx_.Check(xvcgpu_picture_create(ctx_.get(), width, height, bitdepth, &pic_));
  }
  // a view of a picture somebody else owns
  Picture(const Context &ctx, xvcgpu_picture *borrowed)
      : ctx_(ctx), pic_(borrowed), owned_(false) {}
  ~Picture() {
    if (owned_) xvcgpu_picture_destroy(pic_);
  }
  Picture(const Picture &) = delete;
  Picture &operator=(const Picture &) = delete;
  xvcgpu_picture *get() const { return pic_; }
  // planes[c] -> sample (0,0), strides in samples (YuvPicture::GetSamplePtr /
  // GetStride)
  void Upload(const uint16_t *const planes[3], const ptrdiff_t strides[3]) {
    ctx_.Check(xvcgpu_picture_upload(pic_, planes, strides));
  }
  void Download(uint16_t *const planes[3], const ptrdiff_t strides[3]) const {
    ctx_.Check(xvcgpu_picture_download(pic_, planes, strides));
  }
  void PadBorder() { ctx_.Check(xvcgpu_pad_border(ctx_.get(), pic_)); }

 private:
  const Context &ctx_;
  xvcgpu_picture *pic_;
  bool owned_;
};

// SampleMetric(simd, bitdepth, type, structural_strength) ::Compare, batched.
class SampleMetric {
 public:
  SampleMetric(const Context &ctx, int structural_strength = 16)
      : ctx_(ctx), strength_(structural_strength) {}
  // dist[i] = Compare(qp, comp, w, h, src1 block, src2 block displaced by mv)
  std::vector<uint64_t> CompareBatch(const Picture &src1, const Picture &src2,
                                     int comp, double distortion_weight,
                                     const std::vector<xvcgpu_metric_cand> &cands) const {
    DeviceArray<xvcgpu_metric_cand> d(ctx_, cands);
    DeviceArray<uint64_t> out(ctx_, cands.size());
    ctx_.Check(xvcgpu_metric_batch(ctx_.get(), src1.get(), src2.get(), comp,
                                   distortion_weight, strength_, d.data(),
                                   static_cast<int>(cands.size()), out.data()));
    return out.ToHost();
  }

 private:
  const Context &ctx_;
  int strength_;
};

// InterSearch::MotionEstNormal with SearchMethod::kTzSearch, batched over CUs.
class InterSearch {
 public:
  explicit InterSearch(const Context &ctx) : ctx_(ctx) {}
  std::vector<xvcgpu_me_result> MotionEstNormalBatch(
      const Picture &orig_pic, const Picture &ref_pic,
      const std::vector<xvcgpu_me_block> &blocks) const {
    DeviceArray<xvcgpu_me_block> d(ctx_, blocks);
    DeviceArray<xvcgpu_me_result> r(ctx_, blocks.size());
    // CUs that try local illumination compensation search with the AC-only
    // metrics (GetFullpelMetric / GetSubpelMetric): their own kernel instances
    int lic = 0;
    for (size_t i = 0; i < blocks.size(); i++)
      if (blocks[i].fullpel_mv & XVC_ME_USE_LIC) lic = XVCGPU_ME_LIC_JOBS;
    ctx_.Check(xvcgpu_me_search(ctx_.get(), orig_pic.get(), ref_pic.get(),
                                XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL | lic, d.data(),
                                static_cast<int>(blocks.size()), r.data()));
    return r.ToHost();
  }
  // InterSearch::GetSubpelDist per candidate MV (the step EvalStartMvp and
  // SearchMergeCandidates repeat): motion-compensate the luma block, Compare.
  std::vector<uint64_t> GetSubpelDistBatch(
      const Picture &orig_pic, const Picture &ref_pic,
      const std::vector<xvcgpu_mc_metric_cand> &cands,
      int structural_strength = 16) const {
    DeviceArray<xvcgpu_mc_metric_cand> d(ctx_, cands);
    DeviceArray<uint64_t> r(ctx_, cands.size());
    ctx_.Check(xvcgpu_mc_metric_batch(ctx_.get(), orig_pic.get(), ref_pic.get(),
                                      structural_strength, d.data(),
                                      static_cast<int>(cands.size()), r.data()));
    return r.ToHost();
  }
  // ---- the host control around the batches (inter_search.cc:437-578) --------
  // Bits of an MV difference as the search prices it (GetNumExpGolombBits,
  // GetMvdBits, GetMvpBits: inter_search.cc:1139-1190).  Vectors in 1/16 pel,
  // differences coded in 1/4 pel (MvDelta::kPrecisionShift = 2), or in whole
  // samples for fullpel-MV CUs (mvd_down_shift = 2).
  static uint32_t GetNumExpGolombBits(int mvd) {
    uint32_t length = 1;
    uint32_t v = mvd <= 0 ? (static_cast<uint32_t>(-mvd) << 1) + 1 : static_cast<uint32_t>(mvd) << 1;
    while (v != 1) {
      v >>= 1;
      length += 2;
    }
    return length;
  }
  static uint32_t GetMvdBits(int mvp_x, int mvp_y, int mv_x, int mv_y, int mvd_down_shift) {
    const int shift = 2 + mvd_down_shift;
    return GetNumExpGolombBits((mv_x - mvp_x) >> shift) +
           GetNumExpGolombBits((mv_y - mvp_y) >> shift);
  }
  static uint32_t GetMvpBits(int /*mvp_idx*/, int num_mvp) { return num_mvp == 1 ? 0 : 1; }

  // InterSearch::EvalStartMvp (inter_search.cc:966-997) for a batch of CUs with
  // kNumInterMvPredictors = 2 candidates each: SAD of the motion-compensated
  // luma block at each (clipped) predictor + the predictor index bits, first
  // strictly smaller cost wins.  mvp[i] = {x0, y0, x1, y1} in 1/16 pel.
  struct StartMvp {
    int idx;
    uint32_t cost;
  };
  std::vector<StartMvp> EvalStartMvpBatch(const Picture &orig_pic, const Picture &ref_pic,
                                          const std::vector<xvcgpu_me_block> &blocks,
                                          const std::vector<std::array<int32_t, 4>> &mvp) const {
    std::vector<xvcgpu_mc_metric_cand> cands;
    cands.reserve(2 * blocks.size());
    for (size_t i = 0; i < blocks.size(); i++)
      for (int k = 0; k < 2; k++) {
        xvcgpu_mc_metric_cand c = xvcgpu_mc_metric_cand();
        c.x = blocks[i].x;
        c.y = blocks[i].y;
        c.w = blocks[i].w;
        c.h = blocks[i].h;
        // GetMvpMetricType (inter_search.cc:1078-1080): kSad for every CU, also
        // one that tries local illumination compensation
        c.metric = XVC_METRIC_SAD;
        c.mv_x = mvp[i][2 * k];
        c.mv_y = mvp[i][2 * k + 1];
        cands.push_back(c);
      }
    const std::vector<uint64_t> dist = GetSubpelDistBatch(orig_pic, ref_pic, cands);
    std::vector<StartMvp> out(blocks.size());
    for (size_t i = 0; i < blocks.size(); i++) {
      StartMvp best = {0, 0xffffffffu};
      for (int k = 0; k < 2; k++) {
        const uint32_t bits = GetMvpBits(k, 2);
        const uint32_t cost = static_cast<uint32_t>(dist[2 * i + k]) +
                              (static_cast<uint32_t>(bits * blocks[i].lambda16 + 0.5) >> 16);
        if (cost < best.cost) {
          best.cost = cost;
          best.idx = k;
        }
      }
      out[i] = best;
    }
    return out;
  }
  // InterSearch::EvalFinalMvpIdx (inter_search.cc:1000-1019): the predictor that
  // makes the found vector cheapest to code; the start predictor wins a tie.
  static int EvalFinalMvpIdx(const std::array<int32_t, 4> &mvp, int mv_x, int mv_y,
                             int mvp_idx_start, bool fullpel_mv) {
    int best = 0;
    uint32_t best_cost = 0xffffffffu;
    for (int k = 0; k < 2; k++) {
      const uint32_t cost =
          GetMvpBits(k, 2) + GetMvdBits(mvp[2 * k], mvp[2 * k + 1], mv_x, mv_y, fullpel_mv ? 2 : 0);
      if (cost < best_cost || (cost == best_cost && k == mvp_idx_start)) {
        best_cost = cost;
        best = k;
      }
    }
    return best;
  }
  // The per-list loop of InterSearch::SearchRefIdx (uni-prediction, TZ search,
  // inter_search.cc:458-578) over the reference pictures of one list, for a
  // batch of CUs that all use the same `previous` vectors: per reference
  // picture one EvalStartMvp batch, one motion search batch from the chosen
  // predictor, EvalFinalMvpIdx, then cost = dist + ((bits * lambda) >> 16) with
  // the first strictly smaller cost kept.  side_bits[r][i]: what
  // GetInterPredBits adds besides the vector difference and predictor index
  // for CU i with reference r (inter direction, reference index: they come from
  // the caller's entropy-coder state, syntax_writer.cc); mvp[r][i] as above.
  struct UniPredChoice {
    int ref_idx, mvp_idx;
    int mv_x, mv_y;
    uint32_t dist, cost;
  };
  std::vector<UniPredChoice> SearchRefIdxBatch(
      const Picture &orig_pic, const std::vector<const Picture *> &ref_pics,
      const std::vector<xvcgpu_me_block> &blocks,
      const std::vector<std::vector<std::array<int32_t, 4>>> &mvp,
      const std::vector<std::vector<uint32_t>> &side_bits) const {
    std::vector<UniPredChoice> best(blocks.size());
    for (size_t i = 0; i < blocks.size(); i++) {
      UniPredChoice c = {-1, 0, 0, 0, 0, 0xffffffffu};
      best[i] = c;
    }
    for (size_t r = 0; r < ref_pics.size(); r++) {
      const std::vector<StartMvp> start = EvalStartMvpBatch(orig_pic, *ref_pics[r], blocks, mvp[r]);
      std::vector<xvcgpu_me_block> jobs(blocks);
      for (size_t i = 0; i < jobs.size(); i++) {
        jobs[i].mvp_x = mvp[r][i][2 * start[i].idx];
        jobs[i].mvp_y = mvp[r][i][2 * start[i].idx + 1];
      }
      const std::vector<xvcgpu_me_result> res = MotionEstNormalBatch(orig_pic, *ref_pics[r], jobs);
      for (size_t i = 0; i < jobs.size(); i++) {
        const bool fp = (jobs[i].fullpel_mv & XVC_ME_FULLPEL_MV) != 0;
        if (res[i].subpel_dist == XVCGPU_ME_UNSUPPORTED)
          throw Error(XVCGPU_UNSUPPORTED, "SearchRefIdxBatch: a job the device search does not take");
        const int idx = EvalFinalMvpIdx(mvp[r][i], res[i].mv_x, res[i].mv_y, start[i].idx, fp);
        const uint32_t bits = side_bits[r][i] + GetMvpBits(idx, 2) +
                              GetMvdBits(mvp[r][i][2 * idx], mvp[r][i][2 * idx + 1], res[i].mv_x,
                                         res[i].mv_y, fp ? 2 : 0);
        const uint32_t cost =
            res[i].subpel_dist +
            static_cast<uint32_t>((static_cast<uint64_t>(bits) * jobs[i].lambda16) >> 16);
        if (cost < best[i].cost) {
          UniPredChoice c = {static_cast<int>(r), idx, res[i].mv_x, res[i].mv_y,
                             res[i].subpel_dist, cost};
          best[i] = c;
        }
      }
    }
    return best;
  }

  // InterSearch::SearchMergeCandidates (inter_search.cc:165-197) for a batch of
  // CUs: SATD of the prediction every one of the kNumInterMergeCandidates = 5
  // merge candidates gives (uni- or bi-directional; reference index 0 of each
  // list here: the pictures are arguments), cost = dist + bits * sqrt(lambda) in
  // double with bits = idx + 1 (the last index one less), stable sort, and the
  // cut: the first kFastMergeNumCand = 4, fewer if a candidate costs more than
  // 1.25 x the best.  `pred` is scratch for the bi-directional candidates.
  static constexpr int kNumMergeCand = 5;   // constants::kNumInterMergeCandidates (common.h:122)
  struct MergeCand {
    int inter_dir;            // 0: L0, 1: L1, 2: bi
    int mv[2][2];             // [list][x, y], 1/16 pel
  };
  struct MergeChoice {
    int order[kNumMergeCand];            // candidate indices, cheapest first
    int num;                  // how many of them the RD search should try
  };
  std::vector<MergeChoice> SearchMergeCandidatesBatch(
      const Picture &orig_pic, const Picture &ref_l0, const Picture &ref_l1, Picture *pred,
      const std::vector<xvcgpu_me_block> &blocks,
      const std::vector<std::array<MergeCand, kNumMergeCand>> &cands,
      const std::vector<double> &lambda_sqrt) const {
    const size_t n = blocks.size();
    std::vector<std::array<uint64_t, kNumMergeCand>> dist(n);
    for (int m = 0; m < kNumMergeCand; m++) {
      std::vector<xvcgpu_mc_metric_cand> uni[2];
      std::vector<size_t> uni_of[2], bi_of;
      std::vector<xvcgpu_mc_bi_block> bi;
      std::vector<xvcgpu_metric_cand> bi_metric;
      for (size_t i = 0; i < n; i++) {
        const MergeCand &c = cands[i][m];
        if (c.inter_dir == 2) {
          xvcgpu_mc_bi_block b = xvcgpu_mc_bi_block();
          b.x = blocks[i].x; b.y = blocks[i].y; b.w = blocks[i].w; b.h = blocks[i].h;
          b.comp = 0;
          b.mv0_x = c.mv[0][0]; b.mv0_y = c.mv[0][1];
          b.mv1_x = c.mv[1][0]; b.mv1_y = c.mv[1][1];
          bi.push_back(b);
          xvcgpu_metric_cand k = xvcgpu_metric_cand();
          k.x = blocks[i].x; k.y = blocks[i].y; k.w = blocks[i].w; k.h = blocks[i].h;
          k.metric = XVC_METRIC_SATD;
          bi_metric.push_back(k);
          bi_of.push_back(i);
        } else {
          const int l = c.inter_dir;
          xvcgpu_mc_metric_cand k = xvcgpu_mc_metric_cand();
          k.x = blocks[i].x; k.y = blocks[i].y; k.w = blocks[i].w; k.h = blocks[i].h;
          k.metric = XVC_METRIC_SATD;
          k.mv_x = c.mv[l][0]; k.mv_y = c.mv[l][1];
          uni[l].push_back(k);
          uni_of[l].push_back(i);
        }
      }
      for (int l = 0; l < 2; l++)
        if (!uni[l].empty()) {
          const std::vector<uint64_t> d = GetSubpelDistBatch(orig_pic, l ? ref_l1 : ref_l0, uni[l]);
          for (size_t k = 0; k < d.size(); k++) dist[uni_of[l][k]][m] = d[k];
        }
      if (!bi.empty()) {
        // (the blocks of one batch may overlap in `pred` only if they overlap in
        // the picture: the caller's CUs of one size class do not)
        DeviceArray<xvcgpu_mc_bi_block> db(ctx_, bi);
        ctx_.Check(xvcgpu_mc_bipred_batch(ctx_.get(), ref_l0.get(), ref_l1.get(), pred->get(),
                                          db.data(), static_cast<int>(bi.size())));
        DeviceArray<xvcgpu_metric_cand> dm(ctx_, bi_metric);
        DeviceArray<uint64_t> out(ctx_, bi_metric.size());
        ctx_.Check(xvcgpu_metric_batch(ctx_.get(), orig_pic.get(), pred->get(), 0, 1.0, 16,
                                       dm.data(), static_cast<int>(bi_metric.size()),
                                       out.data()));
        const std::vector<uint64_t> d = out.ToHost();
        for (size_t k = 0; k < d.size(); k++) dist[bi_of[k]][m] = d[k];
      }
    }
    std::vector<MergeChoice> out(n);
    for (size_t i = 0; i < n; i++) out[i] = FoldMergeCandidates(dist[i].data(), lambda_sqrt[i]);
    return out;
  }
  // the fold alone (host arithmetic, exactly the reference's doubles)
  static MergeChoice FoldMergeCandidates(const uint64_t dist[kNumMergeCand], double lambda_sqrt) {
    std::pair<int, double> c[kNumMergeCand];
    for (int m = 0; m < kNumMergeCand; m++) {
      const uint32_t bits = static_cast<uint32_t>(m + 1 - (m < kNumMergeCand - 1 ? 0 : 1));
      c[m] = std::make_pair(m, static_cast<double>(dist[m]) + bits * lambda_sqrt);
    }
    std::stable_sort(c, c + kNumMergeCand, [](const std::pair<int, double> &a, const std::pair<int, double> &b) {
      return a.second < b.second;
    });
    MergeChoice r;
    for (int m = 0; m < kNumMergeCand; m++) r.order[m] = c[m].first;
    r.num = 4;
    for (int m = 4; m >= 0; m--)
      if (c[m].second > c[0].second * 1.25) r.num = m;
    return r;
  }
  // SearchMotion's final choice between the best list-0 state, the best list-1
  // state on a reference picture list 0 does not have, and the bi-directional
  // state (inter_search.cc:247-257): 0 = bi, 1 = L0, 2 = L1
  static int ChooseUniOrBi(uint32_t cost_l0, uint32_t cost_l1_unique, uint32_t cost_bi) {
    if (cost_bi <= cost_l0 && cost_bi <= cost_l1_unique) return 0;
    return cost_l0 <= cost_l1_unique ? 1 : 2;
  }

  // One SearchBiIterative refinement step per job (inter_search.cc:392-433):
  // ref_other = picture of the list whose MV is fixed (job.other_mv),
  // ref_search = picture of the list being refined.
  std::vector<xvcgpu_me_result> SearchBiStepBatch(
      const Picture &orig_pic, const Picture &ref_other, const Picture &ref_search,
      const std::vector<xvcgpu_bi_block> &jobs) const {
    DeviceArray<xvcgpu_bi_block> d(ctx_, jobs);
    DeviceArray<xvcgpu_me_result> r(ctx_, jobs.size());
    ctx_.Check(xvcgpu_bipred_search(ctx_.get(), orig_pic.get(), ref_other.get(),
                                    ref_search.get(), d.data(),
                                    static_cast<int>(jobs.size()), r.data(), 64));
    return r.ToHost();
  }
  // InterSearch::SearchBiIterative (inter_search.cc:392-433) around the device
  // steps, for a batch of CUs with one reference picture per list.  Per CU the
  // loop is the reference's: start with the list that lost the uni-directional
  // comparison; predict from the other list with the CU's current vector, search
  // +-4 around the searched list's best uni-directional vector with the predictor
  // that search ended on (SearchRefIdx, bipred branch, :485-489), EvalFinalMvpIdx,
  // remember vector and predictor as the list's new bootstrap when more than one
  // iteration is configured (:550-554), price the pair
  // cost = dist + ((bits * lambda) >> 16), keep it if strictly cheaper, go back to
  // the best state, stop when an iteration did not improve, else swap lists.
  // Every iteration is one SearchBiStepBatch per searched list over the CUs still
  // iterating.  side_bits[i]: GetInterPredBits of the bi-directional CU without
  // the two predictor indices and vector differences (5 with
  // fast_inter_pred_bits, :1108).
  struct ListState {
    int mvp_idx;
    int mv_x, mv_y;
  };
  struct BiPredChoice {
    ListState list[2];
    uint32_t cost;
    int steps;                // device steps this CU took
  };
  std::vector<BiPredChoice> SearchBiIterativeBatch(
      const Picture &orig_pic, const Picture &ref_l0, const Picture &ref_l1,
      const std::vector<xvcgpu_me_block> &blocks,
      const std::vector<std::array<int32_t, 4>> mvp[2], const std::vector<ListState> uni[2],
      const std::vector<int> &best_uni_dir, const std::vector<uint32_t> &side_bits,
      int num_iterations) const {
    const size_t n = blocks.size();
    std::vector<BiPredChoice> best(n);
    std::vector<ListState> boot[2] = {uni[0], uni[1]};
    std::vector<int> search_list(n);
    std::vector<char> active(n, 1);
    for (size_t i = 0; i < n; i++) {
      BiPredChoice c = {{uni[0][i], uni[1][i]}, 0xffffffffu, 0};
      best[i] = c;
      search_list[i] = best_uni_dir[i] == 0 ? 1 : 0;
    }
    const Picture *ref[2] = {&ref_l0, &ref_l1};
    for (int it = 0; it < num_iterations; it++) {
      bool any = false;
      const std::vector<int> searched(search_list);   // one step per CU and iteration
      for (int l = 0; l < 2; l++) {
        std::vector<xvcgpu_bi_block> jobs;
        std::vector<size_t> of;
        for (size_t i = 0; i < n; i++) {
          if (!active[i] || searched[i] != l) continue;
          xvcgpu_bi_block j = xvcgpu_bi_block();
          j.blk = blocks[i];
          j.blk.mvp_x = mvp[l][i][2 * boot[l][i].mvp_idx];
          j.blk.mvp_y = mvp[l][i][2 * boot[l][i].mvp_idx + 1];
          j.other_mv_x = best[i].list[1 - l].mv_x;
          j.other_mv_y = best[i].list[1 - l].mv_y;
          j.boot_mv_x = boot[l][i].mv_x;
          j.boot_mv_y = boot[l][i].mv_y;
          jobs.push_back(j);
          of.push_back(i);
        }
        if (jobs.empty()) continue;
        any = true;
        const std::vector<xvcgpu_me_result> res =
            SearchBiStepBatch(orig_pic, *ref[1 - l], *ref[l], jobs);
        for (size_t k = 0; k < of.size(); k++) {
          const size_t i = of[k];
          const bool fp = (blocks[i].fullpel_mv & XVC_ME_FULLPEL_MV) != 0;
          ListState cand = {EvalFinalMvpIdx(mvp[l][i], res[k].mv_x, res[k].mv_y,
                                            boot[l][i].mvp_idx, fp),
                            res[k].mv_x, res[k].mv_y};
          if (num_iterations > 1) boot[l][i] = cand;
          ListState st[2];
          st[l] = cand;
          st[1 - l] = best[i].list[1 - l];
          uint32_t bits = side_bits[i];
          for (int q = 0; q < 2; q++)
            bits += GetMvpBits(st[q].mvp_idx, 2) +
                    GetMvdBits(mvp[q][i][2 * st[q].mvp_idx], mvp[q][i][2 * st[q].mvp_idx + 1],
                               st[q].mv_x, st[q].mv_y, fp ? 2 : 0);
          const uint32_t cost =
              res[k].subpel_dist +
              static_cast<uint32_t>((static_cast<uint64_t>(bits) * blocks[i].lambda16) >> 16);
          best[i].steps++;
          if (cost < best[i].cost) {
            best[i].cost = cost;
            best[i].list[l] = cand;
            search_list[i] = 1 - l;
          } else {
            active[i] = 0;  // cost_best == prev_best (:427)
          }
        }
      }
      if (!any) break;
    }
    return best;
  }
  // InterSearch::SearchMotion (inter_search.cc:198-259) for a batch of CUs of a
  // bi-predicted picture with one reference per list (both "unique"): the
  // uni-directional searches of both lists, SearchBiIterative, and the choice.
  // side_bits_uni[l][i] / side_bits_bi[i]: what GetInterPredBits adds besides the
  // predictor indices and vector differences.
  struct MotionChoice {
    int inter_dir;            // 0: L0, 1: L1, 2: bi
    ListState list[2];
    uint32_t cost;
    ListState uni[2];         // the lists' uni-directional results
    uint32_t cost_uni[2], cost_bi;
    int bi_steps;
  };
  std::vector<MotionChoice> SearchMotionBatch(
      const Picture &orig_pic, const Picture &ref_l0, const Picture &ref_l1,
      const std::vector<xvcgpu_me_block> blocks[2],
      const std::vector<std::array<int32_t, 4>> mvp[2],
      const std::vector<uint32_t> side_bits_uni[2], const std::vector<uint32_t> &side_bits_bi,
      int num_iterations) const {
    const size_t n = blocks[0].size();
    const Picture *ref[2] = {&ref_l0, &ref_l1};
    std::vector<UniPredChoice> u[2];
    std::vector<ListState> uni[2];
    for (int l = 0; l < 2; l++) {
      u[l] = SearchRefIdxBatch(orig_pic, std::vector<const Picture *>(1, ref[l]), blocks[l],
                               std::vector<std::vector<std::array<int32_t, 4>>>(1, mvp[l]),
                               std::vector<std::vector<uint32_t>>(1, side_bits_uni[l]));
      uni[l].resize(n);
      for (size_t i = 0; i < n; i++) {
        ListState s = {u[l][i].mvp_idx, u[l][i].mv_x, u[l][i].mv_y};
        uni[l][i] = s;
      }
    }
    std::vector<int> best_uni_dir(n);
    for (size_t i = 0; i < n; i++) best_uni_dir[i] = u[0][i].cost <= u[1][i].cost ? 0 : 1;
    const std::vector<BiPredChoice> bi = SearchBiIterativeBatch(
        orig_pic, ref_l0, ref_l1, blocks[0], mvp, uni, best_uni_dir, side_bits_bi, num_iterations);
    std::vector<MotionChoice> out(n);
    for (size_t i = 0; i < n; i++) {
      const int pick = ChooseUniOrBi(u[0][i].cost, u[1][i].cost, bi[i].cost);
      MotionChoice c;
      c.inter_dir = pick == 0 ? 2 : pick - 1;
      c.list[0] = pick == 0 ? bi[i].list[0] : uni[0][i];
      c.list[1] = pick == 0 ? bi[i].list[1] : uni[1][i];
      c.cost = pick == 0 ? bi[i].cost : u[pick - 1][i].cost;
      for (int l = 0; l < 2; l++) {
        c.uni[l] = uni[l][i];
        c.cost_uni[l] = u[l][i].cost;
      }
      c.cost_bi = bi[i].cost;
      c.bi_steps = bi[i].steps;
      out[i] = c;
    }
    return out;
  }
  // ---- SearchMotion as the reference configures itself -----------------------
  // Up to kMaxRefs reference pictures per list (default_num_ref_pics = 2, 3 in
  // placebo: encoder_settings.cc:36, :48), lists that may name the same pictures,
  // pictures with only back references, closed-form bit prices.
  static constexpr int kMaxRefs = 3;
  // The picture-level inputs of SearchMotion (ReferencePictureLists + PictureData):
  struct RefLists {
    int num_ref[2];
    const Picture *pic[2][kMaxRefs];
    // ReferencePictureLists::GetSamePocMappingFor(L1) (reference_picture_lists.cc:
    // 105-122): the list-0 index of the same picture, or -1 (a "unique" picture)
    int same_poc_in_l0[kMaxRefs];
    bool pic_is_uni;           // PicturePredictionType::kUni: inter_dir costs 1 bit, else 3
    bool force_l1_mvd_zero;    // PictureData::GetForceBipredL1MvdZero (only back references)
  };
  // GetInterPredBits with fast_inter_pred_bits (inter_search.cc:1084-1130) without
  // the vector differences: inter direction + reference index (truncated unary: the
  // last index one bit less) + predictor index
  static uint32_t RefIdxBits(int num_ref, int ref_idx) {
    if (num_ref <= 1) return 0;
    return static_cast<uint32_t>(ref_idx + 1 - (ref_idx == num_ref - 1 ? 1 : 0));
  }
  struct ListChoice {
    int ref_idx, mvp_idx;
    int mv_x, mv_y;
  };
  struct MotionChoiceMulti {
    int inter_dir;             // 0: L0, 1: L1, 2: bi
    ListChoice list[2];
    uint32_t cost;
    uint32_t cost_l0, cost_l1, cost_l1_unique, cost_bi;
    ListChoice uni[2];         // the lists' uni-directional choices
    int bi_steps;
    // TzSearch's result per (list, picture) actually searched: the next CU's
    // previous_fullpel_ (inter_search.cc:637); untouched where nothing was searched
    int32_t fullpel[2][kMaxRefs][2];
  };
  // blocks[l][r][i]: CU i's descriptor for reference r of list l (previous vector
  // and search range are per picture); mvp[l][r][i] = {x0, y0, x1, y1}.
  std::vector<MotionChoiceMulti> SearchMotionMultiBatch(
      const Picture &orig_pic, const RefLists &refs,
      const std::vector<xvcgpu_me_block> blocks[2][kMaxRefs],
      const std::vector<std::array<int32_t, 4>> mvp[2][kMaxRefs], int num_iterations) const {
    const size_t n = blocks[0][0].size();
    const uint32_t kNone = 0xffffffffu;
    std::vector<MotionChoiceMulti> out(n);
    // per (list, picture): what SearchRefIdx remembers for the bi-directional
    // refinement (unipred_best_mv_ / _mvp_idx_ / _dist_, :550-554)
    struct Uni {
      int mvp_idx, mv_x, mv_y;
      uint32_t dist;
    };
    std::vector<Uni> uni[2][kMaxRefs];
    std::vector<ListChoice> best[2], best_unique(n);
    std::vector<uint32_t> cost_best[2], cost_unique(n, kNone);
    for (size_t i = 0; i < n; i++)
      for (int l = 0; l < 2; l++)
        for (int r = 0; r < kMaxRefs; r++) out[i].fullpel[l][r][0] = out[i].fullpel[l][r][1] = 0;
    for (int l = 0; l < 2; l++) {
      ListChoice none = {-1, 0, 0, 0};
      best[l].assign(n, none);
      cost_best[l].assign(n, kNone);
      const uint32_t dir_bits = refs.pic_is_uni ? 1 : 3;
      for (int r = 0; r < refs.num_ref[l]; r++) {
        const std::vector<xvcgpu_me_block> &blk = blocks[l][r];
        const std::vector<std::array<int32_t, 4>> &pl = mvp[l][r];
        const bool unique = l == 1 && refs.same_poc_in_l0[r] < 0;
        const bool reuse = l == 1 && !unique;
        const bool force_zero = refs.force_l1_mvd_zero && l == 1;
        const std::vector<StartMvp> start = EvalStartMvpBatch(orig_pic, *refs.pic[l][r], blk, pl);
        uni[l][r].resize(n);
        if (force_zero) {
          // :496-518: the predictor itself is the candidate, priced by EvalStartMvp
          for (size_t i = 0; i < n; i++)
            if (start[i].cost < cost_best[l][i]) {
              ListChoice c = {r, start[i].idx, pl[i][2 * start[i].idx], pl[i][2 * start[i].idx + 1]};
              best[l][i] = c;
              cost_best[l][i] = start[i].cost;
            }
          if (!unique) continue;
        }
        std::vector<xvcgpu_me_result> res;
        if (!reuse) {
          std::vector<xvcgpu_me_block> jobs(blk);
          for (size_t i = 0; i < n; i++) {
            jobs[i].mvp_x = pl[i][2 * start[i].idx];
            jobs[i].mvp_y = pl[i][2 * start[i].idx + 1];
          }
          res = MotionEstNormalBatch(orig_pic, *refs.pic[l][r], jobs);
        }
        for (size_t i = 0; i < n; i++) {
          const bool fp = (blk[i].fullpel_mv & XVC_ME_FULLPEL_MV) != 0;
          int mv_x, mv_y;
          uint32_t dist;
          if (reuse) {                 // :536-542: list 0 searched this picture already
            const Uni &u0 = uni[0][refs.same_poc_in_l0[r]][i];
            mv_x = u0.mv_x;
            mv_y = u0.mv_y;
            dist = u0.dist;
          } else {
            if (res[i].subpel_dist == XVCGPU_ME_UNSUPPORTED)
              throw Error(XVCGPU_UNSUPPORTED, "SearchMotionMultiBatch: job not taken by the search");
            mv_x = res[i].mv_x;
            mv_y = res[i].mv_y;
            dist = res[i].subpel_dist;
            out[i].fullpel[l][r][0] = res[i].fullpel_x;
            out[i].fullpel[l][r][1] = res[i].fullpel_y;
          }
          const int idx = EvalFinalMvpIdx(pl[i], mv_x, mv_y, start[i].idx, fp);
          Uni u = {idx, mv_x, mv_y, dist};
          uni[l][r][i] = u;
          const uint32_t bits = dir_bits + RefIdxBits(refs.num_ref[l], r) + GetMvpBits(idx, 2) +
                                GetMvdBits(pl[i][2 * idx], pl[i][2 * idx + 1], mv_x, mv_y, fp ? 2 : 0);
          const uint32_t cost =
              dist + static_cast<uint32_t>((static_cast<uint64_t>(bits) * blk[i].lambda16) >> 16);
          ListChoice c = {r, idx, mv_x, mv_y};
          if (!force_zero && cost < cost_best[l][i]) {
            cost_best[l][i] = cost;
            best[l][i] = c;
          }
          if (unique && cost < cost_unique[i]) {
            cost_unique[i] = cost;
            best_unique[i] = c;
          }
        }
      }
    }
    // ---- SearchBiIterative (:392-433) over (searched, other) picture pairs -----
    struct BiState {
      ListChoice list[2];
      uint32_t cost;
      int search_list;
      bool active;
    };
    std::vector<BiState> bi(n);
    int iterations = num_iterations;
    for (size_t i = 0; i < n; i++) {
      BiState b = {{best[0][i], best[1][i]}, kNone,
                   cost_best[0][i] <= cost_best[1][i] ? 1 : 0, true};
      if (refs.force_l1_mvd_zero) b.search_list = 0;
      bi[i] = b;
      out[i].bi_steps = 0;
    }
    if (refs.force_l1_mvd_zero) iterations = 1;
    for (int it = 0; it < iterations; it++) {
      bool any = false;
      // this iteration's candidates: per CU every picture of its searched list
      struct Cand {
        size_t cu;
        int r;
      };
      std::vector<std::vector<xvcgpu_bi_block>> jobs(2 * kMaxRefs * kMaxRefs);
      std::vector<std::vector<Cand>> of(2 * kMaxRefs * kMaxRefs);
      for (size_t i = 0; i < n; i++) {
        if (!bi[i].active) continue;
        const int l = bi[i].search_list, o = 1 - l;
        const ListChoice &other = bi[i].list[o];
        for (int r = 0; r < refs.num_ref[l]; r++) {
          const Uni &u = uni[l][r][i];
          xvcgpu_bi_block j = xvcgpu_bi_block();
          j.blk = blocks[l][r][i];
          j.blk.mvp_x = mvp[l][r][i][2 * u.mvp_idx];
          j.blk.mvp_y = mvp[l][r][i][2 * u.mvp_idx + 1];
          j.other_mv_x = other.mv_x;
          j.other_mv_y = other.mv_y;
          j.boot_mv_x = u.mv_x;
          j.boot_mv_y = u.mv_y;
          const int g = (l * kMaxRefs + r) * kMaxRefs + other.ref_idx;
          jobs[g].push_back(j);
          Cand c = {i, r};
          of[g].push_back(c);
        }
      }
      // every candidate of this iteration is evaluated against the state the CU
      // had when the iteration began; the fold below walks them in picture order
      std::vector<std::vector<xvcgpu_me_result>> res(jobs.size());
      for (size_t g = 0; g < jobs.size(); g++) {
        if (jobs[g].empty()) continue;
        any = true;
        const int l = static_cast<int>(g) / (kMaxRefs * kMaxRefs);
        const int r = static_cast<int>(g) / kMaxRefs % kMaxRefs, ro = static_cast<int>(g) % kMaxRefs;
        res[g] = SearchBiStepBatch(orig_pic, *refs.pic[1 - l][ro], *refs.pic[l][r], jobs[g]);
      }
      if (!any) break;
      std::vector<uint32_t> prev_best(n);
      std::vector<BiState> next(bi);
      for (size_t i = 0; i < n; i++) prev_best[i] = bi[i].cost;
      for (int l = 0; l < 2; l++)
        for (int r = 0; r < kMaxRefs; r++)
          for (int ro = 0; ro < kMaxRefs; ro++) {
            const size_t g = static_cast<size_t>((l * kMaxRefs + r) * kMaxRefs + ro);
            for (size_t k = 0; k < of[g].size(); k++) {
              const size_t i = of[g][k].cu;
              const xvcgpu_me_block &blk = blocks[l][r][i];
              const bool fp = (blk.fullpel_mv & XVC_ME_FULLPEL_MV) != 0;
              const std::array<int32_t, 4> &pl = mvp[l][r][i];
              Uni &u = uni[l][r][i];
              const int idx = EvalFinalMvpIdx(pl, res[g][k].mv_x, res[g][k].mv_y, u.mvp_idx, fp);
              ListChoice cand = {r, idx, res[g][k].mv_x, res[g][k].mv_y};
              if (num_iterations > 1) {     // :550-554
                Uni nu = {idx, cand.mv_x, cand.mv_y, res[g][k].subpel_dist};
                u = nu;
              }
              ListChoice st[2];
              st[l] = cand;
              st[1 - l] = bi[i].list[1 - l];
              uint32_t bits = 5;
              for (int q = 0; q < 2; q++) {
                const std::array<int32_t, 4> &pq = mvp[q][st[q].ref_idx][i];
                bits += RefIdxBits(refs.num_ref[q], st[q].ref_idx) + GetMvpBits(st[q].mvp_idx, 2);
                if (refs.force_l1_mvd_zero && q == 1) continue;   // GetForceMvdZero
                bits += GetMvdBits(pq[2 * st[q].mvp_idx], pq[2 * st[q].mvp_idx + 1], st[q].mv_x,
                                   st[q].mv_y, fp ? 2 : 0);
              }
              const uint32_t cost =
                  res[g][k].subpel_dist +
                  static_cast<uint32_t>((static_cast<uint64_t>(bits) * blk.lambda16) >> 16);
              out[i].bi_steps++;
              if (cost < next[i].cost) {
                next[i].cost = cost;
                next[i].list[l] = cand;
              }
            }
          }
      for (size_t i = 0; i < n; i++) {
        if (!bi[i].active) continue;
        bi[i].cost = next[i].cost;
        bi[i].list[0] = next[i].list[0];
        bi[i].list[1] = next[i].list[1];
        if (bi[i].cost == prev_best[i])
          bi[i].active = false;        // :427
        else
          bi[i].search_list = 1 - bi[i].search_list;
      }
    }
    for (size_t i = 0; i < n; i++) {
      MotionChoiceMulti &c = out[i];
      c.cost_l0 = cost_best[0][i];
      c.cost_l1 = cost_best[1][i];
      c.cost_l1_unique = cost_unique[i];
      c.cost_bi = bi[i].cost;
      c.uni[0] = best[0][i];
      c.uni[1] = best[1][i];
      const int pick = ChooseUniOrBi(c.cost_l0, c.cost_l1_unique, c.cost_bi);
      c.inter_dir = pick == 0 ? 2 : pick - 1;
      c.list[0] = pick == 0 ? bi[i].list[0] : best[0][i];
      c.list[1] = pick == 0 ? bi[i].list[1] : (pick == 2 ? best_unique[i] : best[1][i]);
      c.cost = pick == 0 ? c.cost_bi : (pick == 1 ? c.cost_l0 : c.cost_l1_unique);
    }
    return out;
  }

  // InterSearch::MotionEstAffine (inter_search.cc:664-749) per job: the gradient
  // iteration from the affine predictor / bootstrap vector on ref_pic; jobs
  // with XVC_AFFINE_ME_BIPRED search against 2 * orig - the other list's affine
  // prediction from ref_other (pass ref_pic when no job has the flag).
  std::vector<xvcgpu_affine_me_result> MotionEstAffineBatch(
      const Picture &orig_pic, const Picture &ref_pic, const Picture &ref_other,
      const std::vector<xvcgpu_affine_me_block> &jobs) const {
    DeviceArray<xvcgpu_affine_me_block> d(ctx_, jobs);
    DeviceArray<xvcgpu_affine_me_result> r(ctx_, jobs.size());
    ctx_.Check(xvcgpu_affine_me_batch(ctx_.get(), orig_pic.get(), ref_pic.get(),
                                      ref_other.get(), d.data(),
                                      static_cast<int>(jobs.size()), r.data()));
    return r.ToHost();
  }

 private:
  const Context &ctx_;
};

// InterPrediction::MotionCompensationMv for uni-pred CUs, batched.
class InterPrediction {
 public:
  explicit InterPrediction(const Context &ctx) : ctx_(ctx) {}
  void MotionCompensationBatch(const Picture &ref_pic, Picture *pred,
                               const std::vector<xvcgpu_mc_block> &blocks) const {
    DeviceArray<xvcgpu_mc_block> d(ctx_, blocks);
    ctx_.Check(xvcgpu_mc_batch(ctx_.get(), ref_pic.get(), pred->get(), d.data(),
                               static_cast<int>(blocks.size())));
    ctx_.Sync();
  }
  // InterPrediction::MotionCompAffine (uni-pred, Sample output), batched.
  void MotionCompAffineBatch(const Picture &ref_pic, Picture *pred,
                             const std::vector<xvcgpu_mc_affine_block> &blocks) const {
    DeviceArray<xvcgpu_mc_affine_block> d(ctx_, blocks);
    ctx_.Check(xvcgpu_mc_affine_batch(ctx_.get(), ref_pic.get(), pred->get(), d.data(),
                                      static_cast<int>(blocks.size())));
    ctx_.Sync();
  }
  // InterPrediction::MotionCompensation for bi-pred CUs (two lists + AddAvg).
  void MotionCompensationBiBatch(const Picture &ref_l0, const Picture &ref_l1,
                                 Picture *pred,
                                 const std::vector<xvcgpu_mc_bi_block> &blocks) const {
    DeviceArray<xvcgpu_mc_bi_block> d(ctx_, blocks);
    ctx_.Check(xvcgpu_mc_bipred_batch(ctx_.get(), ref_l0.get(), ref_l1.get(),
                                      pred->get(), d.data(),
                                      static_cast<int>(blocks.size())));
    ctx_.Sync();
  }

  // MotionCompensationMv for CUs with GetUseLic(): the prediction with the local
  // illumination model (LocalIlluminationComp / DeriveLicParams) applied;
  // rec_pic = the current reconstruction (neighbouring CUs already there).
  void MotionCompensationLicBatch(const Picture &ref_pic, const Picture &rec_pic,
                                  Picture *pred_pic,
                                  const std::vector<xvcgpu_mc_lic_block> &blocks) const {
    DeviceArray<xvcgpu_mc_lic_block> d(ctx_, blocks);
    ctx_.Check(xvcgpu_mc_lic_batch(ctx_.get(), ref_pic.get(), rec_pic.get(),
                                   pred_pic->get(), d.data(),
                                   static_cast<int>(blocks.size())));
    ctx_.Sync();
  }

 private:
  const Context &ctx_;
};

// TransformEncoder::TransformAndReconstruct (QuantFast), batched; returns the
// per-block non-zero counts (cbf = nnz != 0).
class TransformEncoder {
 public:
  explicit TransformEncoder(const Context &ctx) : ctx_(ctx) {}
  std::vector<int32_t> TransformAndReconstructBatch(
      const Picture &orig_pic, const Picture &pred, Picture *rec_pic,
      const std::vector<xvcgpu_tx_block> &blocks,
      std::vector<int16_t> *levels = nullptr,
      std::vector<uint32_t> *level_offsets = nullptr) const {
    std::vector<uint32_t> off(blocks.size());
    uint32_t total = 0;
    for (size_t i = 0; i < blocks.size(); i++) {
      off[i] = total;
      total += static_cast<uint32_t>(blocks[i].w) * blocks[i].h;
    }
    DeviceArray<xvcgpu_tx_block> d(ctx_, blocks);
    DeviceArray<uint32_t> doff(ctx_, off);
    DeviceArray<int16_t> dlev(ctx_, total ? total : 1);
    DeviceArray<int32_t> dnnz(ctx_, blocks.size());
    ctx_.Check(xvcgpu_residual_batch(ctx_.get(), orig_pic.get(), pred.get(),
                                     rec_pic->get(), d.data(),
                                     static_cast<int>(blocks.size()), dlev.data(),
                                     doff.data(), dnnz.data()));
    if (levels) {
      *levels = dlev.ToHost();
      levels->resize(total);
    }
    if (level_offsets) *level_offsets = off;
    return dnnz.ToHost();
  }

 private:
  const Context &ctx_;
};

// DeblockingFilter(pic_data, rec_pic, beta_offset, tc_offset)::DeblockPicture.
class DeblockingFilter {
 public:
  DeblockingFilter(const Context &ctx, Picture *rec_pic, int beta_offset,
                   int tc_offset, int subblock_size = 4)
      : ctx_(ctx), rec_(rec_pic), beta_(beta_offset), tc_(tc_offset),
        sub_(subblock_size) {}
  void DeblockPicture(const std::vector<xvcgpu_cu_info> &cus,
                      const std::vector<int32_t> &cu_map, int map_stride,
                      bool pic_is_bipred) const {
    DeviceArray<xvcgpu_cu_info> dc(ctx_, cus);
    DeviceArray<int32_t> dm(ctx_, cu_map);
    ctx_.Check(xvcgpu_deblock(ctx_.get(), rec_->get(), dc.data(),
                              static_cast<int>(cus.size()), dm.data(), map_stride,
                              pic_is_bipred ? 1 : 0, beta_, tc_, sub_));
    ctx_.Sync();
  }

 private:
  const Context &ctx_;
  Picture *rec_;
  int beta_, tc_, sub_;
};

// Resampler::ConvertFrom / ConvertTo for the cases without resizing: packed
// planar 4:2:0 bytes <-> the device picture at its internal bit depth.
class Resampler {
 public:
  explicit Resampler(const Context &ctx) : ctx_(ctx) {}
  // src_bytes: Y,U,V back to back, rows tightly packed (xvc_enc_pic_buffer
  // layout); the picture may be larger (internal size rounded up): padded by
  // repetition as CopyFromBytesWithPadding does.
  void ConvertFrom(int src_width, int src_height, int src_bitdepth,
                   const uint8_t *src_bytes, Picture *out_pic) const {
    const size_t n = static_cast<size_t>(src_width) * src_height * 3 / 2 *
                     (src_bitdepth > 8 ? 2 : 1);
    DeviceArray<uint8_t> d(ctx_, n);
    ctx_.Check(xvcgpu_memcpy_h2d(ctx_.get(), d.data(), src_bytes, n));
    ctx_.Check(xvcgpu_picture_import(ctx_.get(), out_pic->get(), d.data(), src_width,
                                     src_height, src_bitdepth));
    ctx_.Sync();
  }
  void ConvertTo(const Picture &src_pic, int out_width, int out_height,
                 int out_bitdepth, bool dither, std::vector<uint8_t> *out_bytes) const {
    const size_t n = static_cast<size_t>(out_width) * out_height * 3 / 2 *
                     (out_bitdepth > 8 ? 2 : 1);
    DeviceArray<uint8_t> d(ctx_, n);
    ctx_.Check(xvcgpu_picture_export(ctx_.get(), src_pic.get(), d.data(), out_width,
                                     out_height, out_bitdepth, dither ? 1 : 0));
    *out_bytes = d.ToHost();
  }

 private:
  const Context &ctx_;
};

// Checksum(Method::kCrc, mode)::HashPicture + GetHash.  (Method::kMd5, the
// reference's default, is a serial chain: it stays on the host.)
class Checksum {
 public:
  enum class Mode { kMinOverhead = 0, kMaxRobust = 1 };
  Checksum(const Context &ctx, Mode mode) : ctx_(ctx), mode_(mode) {}
  void HashPicture(const Picture &pic) {
    DeviceArray<uint8_t> d(ctx_, 8);
    ctx_.Check(xvcgpu_picture_crc(ctx_.get(), pic.get(), static_cast<int>(mode_), d.data()));
    const std::vector<uint8_t> h = d.ToHost();
    hash_.assign(h.begin(), h.begin() + (mode_ == Mode::kMaxRobust ? 6 : 2));
  }
  std::vector<uint8_t> GetHash() const { return hash_; }

 private:
  const Context &ctx_;
  Mode mode_;
  std::vector<uint8_t> hash_;
};

// CuEncoder::CalcDeltaQpFromVariance for all CTUs of a picture at once: the
// device computes the per-CTU block-variance statistic, the host the
// floating-point part (cu_encoder.cc:359-363).
class AdaptiveQp {
 public:
  AdaptiveQp(const Context &ctx, int aqp_strength) : ctx_(ctx), strength_(aqp_strength) {}
  // -> one QP offset per CTU in raster order
  std::vector<int> CalcDeltaQpFromVariance(const Picture &orig_pic, int width, int height,
                                           int bitdepth, int ctu_size) const {
    const size_t nb = static_cast<size_t>((width + 15) / 16) * ((height + 15) / 16);
    const size_t nc = static_cast<size_t>((width + ctu_size - 1) / ctu_size) *
                      ((height + ctu_size - 1) / ctu_size);
    DeviceArray<uint64_t> var16(ctx_, nb), ctu_var(ctx_, nc);
    ctx_.Check(xvcgpu_variance_map(ctx_.get(), orig_pic.get(), var16.data(), ctu_size,
                                   ctu_var.data()));
    const std::vector<uint64_t> v = ctu_var.ToHost();
    std::vector<int> dqp(nc);
    const double k = 1.0 * strength_ / 10;
    for (size_t i = 0; i < nc; i++) {
      const double d = k * (1.5 * std::log(static_cast<double>(v[i])) - 15 - 2 * (bitdepth - 8));
      const int q = static_cast<int>(d);
      dqp[i] = q < -3 ? -3 : (q > 7 ? 7 : q);
    }
    return dqp;
  }

 private:
  const Context &ctx_;
  int strength_;
};

// PictureEncoder::DetermineAllowLic for one reference picture: the histogram
// distance comes from the device, the threshold test is the reference's.
inline bool DetermineAllowLic(const Context &ctx, const Picture &orig_pic,
                              const Picture &ref_orig_pic, int width, int height) {
  DeviceArray<int64_t> d(ctx, 1);
  ctx.Check(xvcgpu_histogram_distance(ctx.get(), orig_pic.get(), ref_orig_pic.get(),
                                      d.data()));
  return d.ToHost()[0] > static_cast<int>(0.06 * width * height);
}

// IntraPrediction::FillReferenceState + Predict for a batch of independent
// blocks (decoder reconstruction of intra CUs whose neighbours are done; the
// encoder's final prediction).
class IntraPrediction {
 public:
  explicit IntraPrediction(const Context &ctx) : ctx_(ctx) {}
  void PredictBatch(const Picture &rec_pic, Picture *pred_pic,
                    const std::vector<xvcgpu_intra_block> &blocks) const {
    DeviceArray<xvcgpu_intra_block> d(ctx_, blocks);
    ctx_.Check(xvcgpu_intra_pred_batch(ctx_.get(), rec_pic.get(), pred_pic->get(), d.data(),
                                       static_cast<int>(blocks.size())));
    ctx_.Sync();
  }

 private:
  const Context &ctx_;
};

// The fast pass of IntraSearch::DetermineSlowIntraModes: SATD of every luma
// mode's prediction, [block][mode]; the caller adds the mode bits and sorts.
class IntraSearch {
 public:
  explicit IntraSearch(const Context &ctx) : ctx_(ctx) {}
  std::vector<uint32_t> SatdAllModesBatch(const Picture &orig_pic, const Picture &rec_pic,
                                          const std::vector<xvcgpu_intra_block> &blocks) const {
    int max_size = 4;
    for (size_t i = 0; i < blocks.size(); i++) {
      if (blocks[i].w > max_size) max_size = blocks[i].w;
      if (blocks[i].h > max_size) max_size = blocks[i].h;
    }
    DeviceArray<xvcgpu_intra_block> d(ctx_, blocks);
    DeviceArray<uint32_t> r(ctx_, blocks.size() * XVC_INTRA_NUM_MODES);
    ctx_.Check(xvcgpu_intra_satd_batch(ctx_.get(), orig_pic.get(), rec_pic.get(), d.data(),
                                       static_cast<int>(blocks.size()), r.data(), max_size));
    return r.ToHost();
  }

 private:
  const Context &ctx_;
};

// Walks a reference-style CU map (anything exposing the PictureData /
// CodingUnit accessors named below; picture_data.h:102-107,
// coding_unit.h:84-285) into the flat device format.  Templated so this header
// never includes reference code.
template <typename PictureDataT, typename CuTreeT, typename CompT, typename ListT,
          typename CornerT>
void ExportCuMap(const PictureDataT &pic_data, CuTreeT tree, CompT luma,
                 CompT chroma, ListT l0, ListT l1, const CornerT corners[4],
                 int width, int height, std::vector<xvcgpu_cu_info> *cus,
                 std::vector<int32_t> *cu_map, int *map_stride) {
  const int mw = (width + 3) / 4, mh = (height + 3) / 4;
  *map_stride = mw;
  cu_map->assign(static_cast<size_t>(mw) * mh, -1);
  cus->clear();
  for (int cy = 0; cy < mh; cy++) {
    for (int cx = 0; cx < mw; cx++) {
      if ((*cu_map)[cy * mw + cx] >= 0) continue;
      const auto *cu = pic_data.GetCuAt(tree, cx * 4, cy * 4);
      if (!cu) continue;
      xvcgpu_cu_info ci = xvcgpu_cu_info();
      ci.x = static_cast<uint16_t>(cu->GetPosX(luma));
      ci.y = static_cast<uint16_t>(cu->GetPosY(luma));
      ci.w = static_cast<uint8_t>(cu->GetWidth(luma));
      ci.h = static_cast<uint8_t>(cu->GetHeight(luma));
      ci.intra = cu->IsIntra();
      ci.cbf_luma = cu->GetCbf(luma);
      ci.qp_y = static_cast<int8_t>(cu->GetQp(luma));
      ci.qp_c = static_cast<int8_t>(cu->GetQp(chroma));
      ci.ref_idx0 = static_cast<int8_t>(cu->GetRefIdx(l0));
      ci.ref_poc[0] = cu->HasMv(l0) ? static_cast<int32_t>(cu->GetRefPoc(l0)) : -1;
      ci.ref_poc[1] = cu->HasMv(l1) ? static_cast<int32_t>(cu->GetRefPoc(l1)) : -1;
      for (int k = 0; k < 4; k++) {
        ci.mv[0][k][0] = cu->GetMv(l0, corners[k]).x;
        ci.mv[0][k][1] = cu->GetMv(l0, corners[k]).y;
        ci.mv[1][k][0] = cu->GetMv(l1, corners[k]).x;
        ci.mv[1][k][1] = cu->GetMv(l1, corners[k]).y;
      }
      const int idx = static_cast<int>(cus->size());
      cus->push_back(ci);
      for (int y = ci.y / 4; y < (ci.y + ci.h) / 4 && y < mh; y++)
        for (int x = ci.x / 4; x < (ci.x + ci.w) / 4 && x < mw; x++)
          (*cu_map)[y * mw + x] = idx;
    }
  }
}

}  // namespace xvc_gpu

#endif  // XVC_AMD_HOST_XVC_GPU_OPS_H_
