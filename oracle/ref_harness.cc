/*
 * ref_harness.cc -- thin extern "C" shim over the REFERENCE's own code.
 *
 * TEST INFRASTRUCTURE ONLY; builds only in the authoring container where
 * /root/reference exists (oracle/Makefile target `ref`), into
 * oracle/_ref/libxvcref.so (git-ignored).  It contains no reference source:
 * it #includes the reference headers (and one .cc, for member templates that
 * are only defined there) from where they lie and calls the reference's
 * classes, so that tests/ can pin the oracle (xvc_oracle.c) against what the
 * reference really computes, and so that tools/gen_golden.py can capture
 * golden vectors.  Each xr_* function has the same signature as the xo_*
 * function it pins.
 */
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <functional>
#include <list>
#include <deque>
#include <sstream>
#include <iostream>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <cstddef>
#include <cstdlib>

#define private public
#define protected public
#include "xvc_common_lib/checksum.h"
#include "xvc_common_lib/coding_unit.h"
#include "xvc_common_lib/deblocking_filter.h"
#include "xvc_common_lib/inter_prediction.h"
#include "xvc_common_lib/intra_prediction.h"
#include "xvc_common_lib/picture_data.h"
#include "xvc_common_lib/quantize.h"
#include "xvc_common_lib/resample.h"
#include "xvc_common_lib/restrictions.h"
#include "xvc_common_lib/segment_header.h"
#include "xvc_common_lib/transform.h"
#include "xvc_common_lib/yuv_pic.h"
#include "xvc_enc_lib/cu_encoder.h"
#include "xvc_enc_lib/encoder_settings.h"
#include "xvc_enc_lib/encoder_simd_functions.h"
#include "xvc_enc_lib/inter_tz_search.h"
#include "xvc_enc_lib/picture_encoder.h"
#include "xvc_enc_lib/rdo_quant.h"
#include "xvc_enc_lib/sample_metric.h"
/* Observation hook for the uni-directional motion search of a real encoder run
 * (tools/gen_me_golden.py): InterSearch::MotionEstNormal constructs a TzSearch
 * and calls Search() on it (inter_search.cc:631-637); while inter_search.cc is
 * compiled below the name resolves to this wrapper, which forwards to the
 * reference's TzSearch and - when capturing - records the call's inputs, its
 * full-pel result, and the sub-pel result the reference's own SubpelSearch /
 * GetSubpelDist give from there (what MotionEstNormal does next, :645-657). */
namespace xvc {
class ObservedTzSearch;
}
/* round 4: one counter over ALL capture tables (tools/gen_order_golden.py): the
 * order in which the encoder's RD search issued the calls, across tables. */
namespace xr_seq {
extern uint32_t g_next;
extern std::vector<uint32_t> g_of[9];   /* 0 me calls, 1 steps, 2 merges, 3 evals, 4 calls,
                                          * 5 cands, 6 finals; round 5: 7 intra SATD calls
                                          * (xr_intra::Call), 8 intra transform calls (IntraTx) */
inline void Stamp(int table) { g_of[table].push_back(g_next++); }
}  // namespace xr_seq
namespace xr_me {
struct Call {
  int32_t poc, ref_poc;
  int16_t x, y;
  uint8_t w, h, depth_nonzero, fullpel_mv;
  int32_t use_lic;   /* AC-only metrics (GetFullpelMetric / GetSubpelMetric) */
  int32_t mvp_x, mvp_y, prev_x, prev_y;
  uint32_t lambda16;
  int32_t search_range;
  int32_t fullpel_x, fullpel_y, mv_x, mv_y;
  uint32_t dist;
};
extern bool g_capture;
extern int g_only_poc;
extern std::set<int> g_also_poc;
extern std::vector<Call> g_calls;
}  // namespace xr_me
namespace xvc {
class ObservedTzSearch {
public:
  ObservedTzSearch(const YuvPicture &orig_pic, const InterPrediction &inter_pred,
                   const EncoderSettings &encoder_settings, int search_range)
    : real_(orig_pic, inter_pred, encoder_settings, search_range),
    orig_pic_(orig_pic), inter_pred_(inter_pred), search_range_(search_range) {
  }
  MvFullpel Search(const CodingUnit &cu, const Qp &qp, const SampleMetric &metric,
                   const MotionVector &mvp, const YuvPicture &ref_pic,
                   const MvFullpel &mv_min, const MvFullpel &mv_max,
                   const MvFullpel &prev_search);
private:
  TzSearch real_;
  const YuvPicture &orig_pic_;
  const InterPrediction &inter_pred_;
  int search_range_;
};
}  // namespace xvc
/* Observation hooks for the REST of the RD search (tools/gen_rd_golden.py):
 *  - SearchRefIdx (inter_search.cc:456-578) stores every search result with
 *    cu->SetMvpIdx(mvp_idx, ref_list) (:556); with the function-like macro below
 *    that statement is followed by xr_rd::AfterMotionEst(...), which sees the
 *    loop's locals: the vector and distortion MotionEstimation returned, the
 *    predictor list, the bootstrap vector.  It records the bi-prediction
 *    refinement steps (FullSearch + sub-pel on the 2 * orig - other target) and
 *    the affine searches (MotionEstAffine, uni and bi).  The one other
 *    SetMvpIdx statement of the file (:510, the force_mvd_zero branch) has no
 *    `mv` / `dist` in scope: there the names resolve to the two XrNone dummies
 *    declared below and the overload for them does nothing.
 *  - SearchMergeCandidates (:165-197) sorts its five candidates' costs with
 *    std::stable_sort (:184); the macro appends xr_rd::AfterMergeSort(...),
 *    which sees the candidate list and the sorted (index, cost) pairs.
 *  - TransformEncoder::TransformAndReconstruct (transform_encoder.cc:203-285) is
 *    compiled below as part of this file with two more macros (see there).
 * Nothing of the reference's control flow is restated: the encoder runs its own
 * code, the hooks only read. */
#include "xvc_enc_lib/cu_writer.h"
namespace xvc {
struct XrNone {};
static const XrNone mv = XrNone(), dist = XrNone();
}  // namespace xvc
namespace xr_rd {
template <typename MV, size_t N>
void AfterMotionEst(xvc::InterSearch *is, xvc::CodingUnit *cu, const xvc::Qp &qp,
                    xvc::RefPicList ref_list, int ref_idx, bool bipred,
                    const std::array<MV, N> &mvp_list, const MV *boot, const MV &mv,
                    xvc::Distortion dist, const xvc::SyntaxWriter &writer, int mvp_idx);
template <typename MV, size_t N>
inline void AfterMotionEst(xvc::InterSearch *, xvc::CodingUnit *, const xvc::Qp &,
                           xvc::RefPicList, int, bool, const std::array<MV, N> &, const MV *,
                           const xvc::XrNone &, const xvc::XrNone &, const xvc::SyntaxWriter &,
                           int) {}
/* round 4: the END of SearchMotion (inter_search.cc:247-257) - its three-way choice
 * loads one of three saved states with cu->LoadStateFrom(state_bi / state_l0 /
 * state_l1_unique_poc); the macro below follows every LoadStateFrom statement of
 * the file with this hook, which looks at the argument's spelling and records the
 * CU's motion state after those three. */
void AfterLoadState(xvc::InterSearch *is, xvc::CodingUnit *cu, const char *what);
void AfterMergeSort(xvc::InterSearch *is, xvc::CodingUnit *cu, const xvc::Qp &qp,
                    const xvc::InterMergeCandidateList &merge_list,
                    const std::array<std::pair<int, double>, 5> &cand_cost);
void AfterQuantRdo(xvc::TransformEncoder *te, xvc::CodingUnit *cu, xvc::YuvComponent comp,
                   const xvc::Qp &qp, const xvc::SyntaxWriter &writer, int non_zero,
                   const xvc::YuvPicture &rec_pic, const xvc::SampleBuffer &pred_buffer);
xvc::Distortion AfterCompare(xvc::TransformEncoder *te, xvc::CodingUnit *cu,
                             xvc::YuvComponent comp, const xvc::YuvPicture &orig_pic,
                             const xvc::SampleBuffer &buffer, const char *func);
uint32_t Crc32(uint32_t crc, const void *data, size_t n);
}  // namespace xr_rd
#define SetMvpIdx(i, l) \
  SetMvpIdx(i, l);      \
  xr_rd::AfterMotionEst(this, cu, qp, ref_list, ref_idx, bipred, mvp_list, mv_bootstrap, mv, dist, \
                        bitstream_writer, mvp_idx)
#define LoadStateFrom(...)      \
  LoadStateFrom(__VA_ARGS__);   \
  xr_rd::AfterLoadState(this, cu, #__VA_ARGS__)
#define stable_sort(a, b, c) \
  stable_sort(a, b, c);      \
  xr_rd::AfterMergeSort(this, cu, qp, merge_list, cand_cost)
#define TzSearch ObservedTzSearch
/* member templates (SubpelSearch<>, ...) are defined only in the .cc */
#include "xvc_enc_lib/inter_search.cc"
#undef TzSearch
#undef SetMvpIdx
#undef stable_sort
#undef LoadStateFrom
/* TransformEncoder::TransformAndReconstruct: after its quantiser call
 * (fwd_quant_.QuantRdo(...), transform_encoder.cc:231-234) the hook records the
 * call - CU, component, transform choice, the CU's motion state, the context
 * states the quantiser read - and the levels it produced; the distortion
 * comparison that ends the function (:284) completes the record with the
 * reconstruction.  CompressAndEvalTransform's cbf-zero comparison (:116-117)
 * goes through the same macro and is recorded as the component's dist_zero. */
#define QuantRdo(a, b, c, d, e, f, g, h, i) \
  QuantRdo(a, b, c, d, e, f, g, h, i);      \
  xr_rd::AfterQuantRdo(this, cu, comp, qp, syntax_writer, non_zero, *rec_pic, pred_buffer)
#define CompareSample(a, b, c, d) \
  CompareSample(a, b, c, d) + xr_rd::AfterCompare(this, cu, comp, c, d, __func__)
#include "xvc_enc_lib/transform_encoder.cc"
#undef QuantRdo
#undef CompareSample
/* IntraPrediction::NeighborState (argument of ComputeRefSamples) likewise */
#include "xvc_common_lib/intra_prediction.cc"
/* IntraSearch::DetermineSlowIntraModes (intra_search.cc:188-305), the SATD
 * pre-selection of the intra search (tools/gen_intra_golden.py): the statement
 * that fetches the predictor modes (:219, once per call) is followed by a hook
 * that records the CU, what DetermineNeighbors says about its surroundings and
 * the reconstruction's row above / column to the left at that moment of the RD
 * search; every evaluated mode's cost statement (:237, :277) ends in a hook that
 * records (mode, SATD).  The hooks only read. */
namespace xr_intra {
void OnCu(xvc::IntraSearch *is, xvc::CodingUnit *cu, const xvc::YuvPicture &rec_pic);
double OnEval(xvc::CodingUnit *cu, int mode, uint64_t dist);
}  // namespace xr_intra
#define GetPredictorLuma(x) \
  GetPredictorLuma(x);      \
  xr_intra::OnCu(this, cu, *rec_pic)
#define GetLambdaSqrt() \
  GetLambdaSqrt() + xr_intra::OnEval(cu, static_cast<int>(intra_mode), dist)
#include "xvc_enc_lib/intra_search.cc"
#undef GetPredictorLuma
#undef GetLambdaSqrt
#undef private
#undef protected

#include "../include/xvcgpu_types.h"
#include "xvc_oracle.h" /* xo_frame_args: the frame pass argument block */

using namespace xvc;  // NOLINT

namespace {

int g_use_simd = 1;

const EncoderSimdFunctions &Simd(int bd) {
  static std::map<int, std::unique_ptr<EncoderSimdFunctions>> cache;
  int key = bd * 2 + g_use_simd;
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::set<CpuCapability> caps;
    if (g_use_simd) caps = SimdCpu::GetRuntimeCapabilities();
    it = cache.emplace(key, std::unique_ptr<EncoderSimdFunctions>(
                                new EncoderSimdFunctions(caps, bd))).first;
  }
  return *it->second;
}

Qp MakeQp(int qp_raw, int bd, double lambda = 1.0) {
  return Qp(qp_raw, ChromaFormat::k420, bd, lambda, 1, 0, 0);
}

/* Copy padded planes (pointer at sample (0,0), >= 80/40 samples of border
 * available on every side) into a reference YuvPicture incl. its border. */
void FillPic(YuvPicture *pic, const uint16_t *const planes[3],
             const ptrdiff_t strides[3]) {
  for (int c = 0; c < 3; c++) {
    YuvComponent comp = YuvComponent(c);
    const int w = pic->GetWidth(comp), h = pic->GetHeight(comp);
    const int bx = static_cast<int>((pic->GetStride(comp) - w) >> 1);
    const int by = bx;
    if (!planes[c]) continue;
    for (int y = -by; y < h + by; y++) {
      Sample *dst = pic->GetSamplePtr(comp, -bx, y);
      const uint16_t *src = planes[c] + y * strides[c] - bx;
      std::memcpy(dst, src, sizeof(Sample) * (w + 2 * bx));
    }
  }
}

void ReadPic(const YuvPicture &pic, uint16_t *const planes[3],
             const ptrdiff_t strides[3], bool with_border) {
  for (int c = 0; c < 3; c++) {
    YuvComponent comp = YuvComponent(c);
    const int w = pic.GetWidth(comp), h = pic.GetHeight(comp);
    const int bx =
        with_border ? static_cast<int>((pic.GetStride(comp) - w) >> 1) : 0;
    if (!planes[c]) continue;
    for (int y = -bx; y < h + bx; y++) {
      const Sample *src = pic.GetSamplePtr(comp, -bx, y);
      uint16_t *dst = planes[c] + y * strides[c] - bx;
      std::memcpy(dst, src, sizeof(Sample) * (w + 2 * bx));
    }
  }
}

}  // namespace

extern "C" {

void xr_set_simd(int use_simd) { g_use_simd = use_simd ? 1 : 0; }

/* ---- metrics ---- */
uint64_t xr_metric_ss(int metric, int bd, int qp_raw_y, int strength,
                      double weight, int w, int h, const uint16_t *s1,
                      ptrdiff_t st1, const uint16_t *s2, ptrdiff_t st2) {
  (void)weight;
  SampleMetric m(Simd(bd).sample_metric, bd, static_cast<MetricType>(metric),
                 strength);
  Qp qp = MakeQp(qp_raw_y, bd);
  return m.CompareSample(qp, YuvComponent::kY, w, h, s1, st1, s2, st2);
}

uint64_t xr_metric_rs(int metric, int bd, int qp_raw_y, int strength,
                      double weight, int w, int h, const int16_t *s1,
                      ptrdiff_t st1, const uint16_t *s2, ptrdiff_t st2) {
  (void)weight;
  SampleMetric m(Simd(bd).sample_metric, bd, static_cast<MetricType>(metric),
                 strength);
  Qp qp = MakeQp(qp_raw_y, bd);
  return m.CompareSample(qp, YuvComponent::kY, w, h, s1, st1, s2, st2);
}

uint64_t xr_ssd_rr(int bd, double weight, int w, int h, const int16_t *s1,
                   ptrdiff_t st1, const int16_t *s2, ptrdiff_t st2) {
  (void)weight;
  SampleMetric m(Simd(bd).sample_metric, bd, MetricType::kSsd);
  Qp qp = MakeQp(32, bd);
  return m.CompareShort(qp, YuvComponent::kY, w, h, s1, st1, s2, st2);
}

uint64_t xr_picture_ssd(int bd, int w, int h, const uint16_t *p1,
                        ptrdiff_t st1, const uint16_t *p2, ptrdiff_t st2,
                        uint64_t *psnr_dist, uint64_t *psnr_samples) {
  /* luma-only pictures of size w x h (chroma planes unused) */
  YuvPicture a(ChromaFormat::k420, w, h, bd, true, 0, 0);
  YuvPicture b(ChromaFormat::k420, w, h, bd, true, 0, 0);
  for (int y = 0; y < h; y++) {
    std::memcpy(a.GetSamplePtr(YuvComponent::kY, 0, y), p1 + y * st1,
                sizeof(Sample) * w);
    std::memcpy(b.GetSamplePtr(YuvComponent::kY, 0, y), p2 + y * st2,
                sizeof(Sample) * w);
  }
  SampleMetric m(Simd(bd).sample_metric, bd, MetricType::kSsd);
  Qp qp = MakeQp(32, bd);
  uint64_t dist =
      m.ComparePicture(qp, YuvComponent::kY, YuvComponent::kY, a, b);
  if (psnr_dist) *psnr_dist = dist;
  if (psnr_samples) {
    /* recover the sample count from the PSNR the reference reports */
    double psnr = m.ComputePsnr(qp, YuvComponent::kY, YuvComponent::kY, a, b);
    if (dist == 0) {
      *psnr_samples = 0;
    } else {
      double mse = 255.0 * 255.0 / std::pow(10.0, psnr / 10.0);
      *psnr_samples = static_cast<uint64_t>(std::llround(dist / mse));
    }
  }
  return dist;
}

/* ---- interpolation ---- */
static InterPrediction *MakeIp(int bd, std::unique_ptr<YuvPicture> *rec) {
  rec->reset(new YuvPicture(ChromaFormat::k420, 8, 8, bd, true, 0, 0));
  return new InterPrediction(Simd(bd).inter_prediction, **rec, bd);
}

void xr_mc_uni(int bd, int is_chroma, int w, int h, int frac_x, int frac_y,
               const uint16_t *ref, ptrdiff_t rs, uint16_t *pred,
               ptrdiff_t ps) {
  std::unique_ptr<YuvPicture> rec;
  std::unique_ptr<InterPrediction> ip(MakeIp(bd, &rec));
  SampleBufferConst rb(ref, rs);
  SampleBuffer pb(pred, ps);
  ip->MotionCompUniPred(w, h, is_chroma ? YuvComponent::kU : YuvComponent::kY,
                        rb, frac_x, frac_y, &pb);
}

void xr_mc_uni_bipred(int bd, int is_chroma, int w, int h, int frac_x,
                      int frac_y, const uint16_t *ref, ptrdiff_t rs,
                      int16_t *pred, ptrdiff_t ps) {
  std::unique_ptr<YuvPicture> rec;
  std::unique_ptr<InterPrediction> ip(MakeIp(bd, &rec));
  SampleBufferConst rb(ref, rs);
  DataBuffer<int16_t> pb(pred, ps);
  ip->MotionCompUniPred(w, h, is_chroma ? YuvComponent::kU : YuvComponent::kY,
                        rb, frac_x, frac_y, &pb);
}

void xr_add_avg(int bd, int w, int h, const int16_t *s1, ptrdiff_t st1,
                const int16_t *s2, ptrdiff_t st2, uint16_t *dst,
                ptrdiff_t ds) {
  /* AddAvgBi body, inter_prediction.cc:1540-1553, via the fn table */
  const int shift = std::max(2, InterPrediction::kInternalPrecision - bd) + 1;
  const int offset = (1 << (shift - 1)) + 2 * InterPrediction::kInternalOffset;
  Simd(bd).inter_prediction.add_avg[w > 2](w, h, offset, shift, bd, s1, st1, s2,
                                           st2, dst, ds);
}

void xr_mc_block(int bd, int comp, int x, int y, int w, int h, int mv_x,
                 int mv_y, int pic_w, int pic_h, const uint16_t *ref_plane,
                 ptrdiff_t rs, uint16_t *pred, ptrdiff_t ps) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  YuvPicture ref_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  const uint16_t *planes[3] = {nullptr, nullptr, nullptr};
  ptrdiff_t strides[3] = {0, 0, 0};
  planes[comp] = ref_plane;
  strides[comp] = rs;
  FillPic(&ref_pic, planes, strides);
  InterPrediction ip(Simd(bd).inter_prediction, ref_pic, bd);
  SampleBuffer pb(pred, ps);
  ip.MotionCompensationMv(*cu, YuvComponent(comp), ref_pic,
                          MotionVector(mv_x, mv_y), false, &pb);
}

void xr_mc_affine_block(int bd, int comp, int x, int y, int w, int h,
                        const int mv[3][2], int pic_w, int pic_h,
                        const uint16_t *ref_plane, ptrdiff_t rs, uint16_t *pred,
                        ptrdiff_t ps) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  YuvPicture ref_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  const uint16_t *planes[3] = {nullptr, nullptr, nullptr};
  ptrdiff_t strides[3] = {0, 0, 0};
  planes[comp] = ref_plane;
  strides[comp] = rs;
  FillPic(&ref_pic, planes, strides);
  InterPrediction ip(Simd(bd).inter_prediction, ref_pic, bd);
  SampleBuffer pb(pred, ps);
  MotionVector3 mv3;
  for (int i = 0; i < 3; i++) mv3[i] = MotionVector(mv[i][0], mv[i][1]);
  ip.MotionCompAffine(*cu, YuvComponent(comp), ref_pic, mv3, &pb);
}

void xr_clip_mv(int pos_x, int pos_y, int pic_w, int pic_h, int *mv_x,
                int *mv_y) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, 8);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, pos_x, pos_y, 8, 8);
  YuvPicture ref_pic(ChromaFormat::k420, pic_w, pic_h, 8, false, 0, 0);
  InterPrediction ip(Simd(8).inter_prediction, ref_pic, 8);
  MotionVector mv(*mv_x, *mv_y);
  ip.ClipMv(*cu, ref_pic, &mv);
  *mv_x = mv.x;
  *mv_y = mv.y;
}

/* ---- transforms ---- */
void xr_fwd_transform(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                      const int16_t *resi, ptrdiff_t rs, int16_t *coeff,
                      ptrdiff_t cs) {
  PictureData pic_data(ChromaFormat::k420, 64, 64, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, 0, 0, w, h);
  cu->SetPredMode(dst4x4 ? PredictionMode::kIntra : PredictionMode::kInter);
  cu->SetTransformType(YuvComponent::kY, static_cast<TransformType>(tx_ver),
                       static_cast<TransformType>(tx_hor));
  ForwardTransform fwd(bd);
  ResidualBuffer in(const_cast<int16_t *>(resi), rs);
  CoeffBuffer out(coeff, cs);
  fwd.Transform(*cu, YuvComponent::kY, in, &out);
}

void xr_inv_transform(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                      int dc_only, const int16_t *coeff, ptrdiff_t cs,
                      int16_t *resi, ptrdiff_t rs) {
  PictureData pic_data(ChromaFormat::k420, 64, 64, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, 0, 0, w, h);
  cu->SetPredMode(dst4x4 ? PredictionMode::kIntra : PredictionMode::kInter);
  cu->SetTransformType(YuvComponent::kY, static_cast<TransformType>(tx_ver),
                       static_cast<TransformType>(tx_hor));
  cu->SetDcCoeffOnly(YuvComponent::kY, dc_only != 0);
  InverseTransform inv(bd);
  CoeffBuffer in(const_cast<int16_t *>(coeff), cs);
  ResidualBuffer out(resi, rs);
  inv.Transform(*cu, YuvComponent::kY, in, &out);
}

/* The same two under Restrictions::disable_ext2_transform_high_precision
 * (restricted mode): thread-local state of the reference, flipped around the
 * call.  tx_hor / tx_ver are the reference's TransformType values (0..5). */
void xr_fwd_transform_restricted(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                                 const int16_t *resi, ptrdiff_t rs, int16_t *coeff,
                                 ptrdiff_t cs) {
  Restrictions &r = Restrictions::GetRW();
  const bool saved = r.disable_ext2_transform_high_precision;
  r.disable_ext2_transform_high_precision = true;
  xr_fwd_transform(bd, w, h, tx_hor, tx_ver, dst4x4, resi, rs, coeff, cs);
  r.disable_ext2_transform_high_precision = saved;
}

void xr_inv_transform_restricted(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                                 int dc_only, const int16_t *coeff, ptrdiff_t cs,
                                 int16_t *resi, ptrdiff_t rs) {
  Restrictions &r = Restrictions::GetRW();
  const bool saved = r.disable_ext2_transform_high_precision;
  r.disable_ext2_transform_high_precision = true;
  xr_inv_transform(bd, w, h, tx_hor, tx_ver, dst4x4, dc_only, coeff, cs, resi, rs);
  r.disable_ext2_transform_high_precision = saved;
}

void xr_fwd_transform_skip(int bd, int w, int h, const int16_t *resi,
                           ptrdiff_t rs, int16_t *coeff, ptrdiff_t cs) {
  ForwardTransform fwd(bd);
  ResidualBuffer in(const_cast<int16_t *>(resi), rs);
  CoeffBuffer out(coeff, cs);
  fwd.TransformSkip(w, h, in, &out);
}

void xr_inv_transform_skip(int bd, int w, int h, const int16_t *coeff,
                           ptrdiff_t cs, int16_t *resi, ptrdiff_t rs) {
  InverseTransform inv(bd);
  CoeffBuffer in(const_cast<int16_t *>(coeff), cs);
  ResidualBuffer out(resi, rs);
  inv.TransformSkip(w, h, in, &out);
}

void xr_dequant(int bd, int qp_raw, int w, int h, const int16_t *in,
                ptrdiff_t is, int16_t *out, ptrdiff_t os) {
  Quantize q;
  Qp qp = MakeQp(qp_raw, bd);
  q.Inverse(YuvComponent::kY, qp, w, h, bd, in, is, out, os);
}

int xr_quant_fast(int bd, int qp_raw, int intra_pic, int w, int h,
                  const int16_t *in, ptrdiff_t is, int16_t *out,
                  ptrdiff_t os) {
  /* QuantFast with sign hiding off: the restriction flag is thread-local
   * state of the reference; flip it around the call. */
  PictureData pic_data(ChromaFormat::k420, 64, 64, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, 0, 0, w, h);
  cu->SetPredMode(PredictionMode::kInter);
  Qp qp = MakeQp(qp_raw, bd);
  EncoderSettings es;
  es.Initialize(SpeedMode::kSlow);
  RdoQuant rq(bd, es);
  Restrictions &r = Restrictions::GetRW();
  bool saved = r.disable_transform_sign_hiding;
  r.disable_transform_sign_hiding = true;
  int nnz = rq.QuantFast(*cu, YuvComponent::kY, qp,
                         intra_pic ? PicturePredictionType::kIntra
                                   : PicturePredictionType::kBi,
                         in, is, out, os);
  r.disable_transform_sign_hiding = saved;
  return nnz;
}

/* ---- transform tables ---- */
const int16_t *xr_transform_matrix(int tx_type, int size) {
  typedef TransformData T;
  switch (tx_type) {
    case XVC_TX_DEFAULT:
    case XVC_TX_DCT2:
      switch (size) {
        case 2: return &T::kDct2Transform2High[0][0];
        case 4: return &T::kDct2Transform4High[0][0];
        case 8: return &T::kDct2Transform8High[0][0];
        case 16: return &T::kDct2Transform16High[0][0];
        case 32: return &T::kDct2Transform32High[0][0];
        case 64: return &T::kDct2Transform64High[0][0];
      }
      return nullptr;
#define XR_TAB(NAME)                                  \
  switch (size) {                                     \
    case 4: return T::k##NAME##Transform4High;        \
    case 8: return T::k##NAME##Transform8High;        \
    case 16: return T::k##NAME##Transform16High;      \
    case 32: return T::k##NAME##Transform32High;      \
    case 64: return T::k##NAME##Transform64High;      \
  }                                                   \
  return nullptr;
    case XVC_TX_DCT2_LOW:
      switch (size) {
        case 4: return &T::kDct2Transform4[0][0];
        case 8: return &T::kDct2Transform8[0][0];
        case 16: return &T::kDct2Transform16[0][0];
        case 32: return &T::kDct2Transform32[0][0];
      }
      return nullptr;
    case XVC_TX_DCT5: XR_TAB(Dct5)
    case XVC_TX_DCT8: XR_TAB(Dct8)
    case XVC_TX_DST1: XR_TAB(Dst1)
    case XVC_TX_DST7: XR_TAB(Dst7)
#undef XR_TAB
  }
  return nullptr;
}

/* ---- deblocking ---- */
void xr_deblock_picture(int bd, int pic_w, int pic_h, int bipred,
                        int beta_offset, int tc_offset, int subblock_size,
                        const xvcgpu_cu_info *cus, int n_cus,
                        uint16_t *const planes[3], const ptrdiff_t strides[3],
                        const int32_t *l0_pocs, int n0, const int32_t *l1_pocs,
                        int n1) {
  (void)subblock_size; /* reference default: 4 */
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  pic_data.SetNalType(bipred ? NalUnitType::kBipredictedPicture
                             : NalUnitType::kPredictedPicture);
  pic_data.SetPoc(1000);
  ReferencePictureLists *rpl = pic_data.GetRefPicLists();
  rpl->Reset(1000);
  auto dummy = std::make_shared<PictureData>(ChromaFormat::k420, 8, 8, bd);
  dummy->SetNalType(NalUnitType::kPredictedPicture);
  for (int i = 0; i < n0; i++)
    rpl->SetRefPic(RefPicList::kL0, i, static_cast<PicNum>(l0_pocs[i]), dummy,
                   nullptr, nullptr);
  for (int i = 0; i < n1; i++)
    rpl->SetRefPic(RefPicList::kL1, i, static_cast<PicNum>(l1_pocs[i]), dummy,
                   nullptr, nullptr);
  SegmentHeader segment;
  segment.chroma_qp_offset_table = 1;
  segment.chroma_qp_offset_u = 0;
  segment.chroma_qp_offset_v = 0;
  segment.max_binary_split_depth = 3;
  Qp pic_qp = MakeQp(32, bd, 57.9);
  pic_data.Init(segment, pic_qp, true);

  for (int i = 0; i < n_cus; i++) {
    const xvcgpu_cu_info &ci = cus[i];
    CodingUnit *cu =
        pic_data.CreateCu(CuTree::Primary, 1, ci.x, ci.y, ci.w, ci.h);
    assert(cu);
    cu->SetSplit(SplitType::kNone);
    cu->SetPredMode(ci.intra ? PredictionMode::kIntra : PredictionMode::kInter);
    cu->SetCbf(YuvComponent::kY, ci.cbf_luma != 0);
    cu->SetQp(ci.qp_y);
    bool use0 = ci.ref_poc[0] >= 0, use1 = ci.ref_poc[1] >= 0;
    cu->SetInterDir(use0 && use1 ? InterDir::kBi
                                 : (use1 ? InterDir::kL1 : InterDir::kL0));
    int idx0 = ci.ref_idx0, idx1 = 0;
    if (use1) {
      for (int k = 0; k < n1; k++)
        if (l1_pocs[k] == ci.ref_poc[1]) { idx1 = k; break; }
    }
    cu->SetRefIdx(idx0, RefPicList::kL0);
    cu->SetRefIdx(idx1, RefPicList::kL1);
    for (int l = 0; l < 2; l++)
      for (int c = 0; c < 4; c++)
        cu->inter_.mv[l][c] = MotionVector(ci.mv[l][c][0], ci.mv[l][c][1]);
    pic_data.MarkUsedInPic(cu);
  }

  YuvPicture rec(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  FillPic(&rec, const_cast<const uint16_t *const *>(planes), strides);
  DeblockingFilter df(&pic_data, &rec, beta_offset, tc_offset);
  df.DeblockPicture();
  ReadPic(rec, planes, strides, false);
}

/* ---- border extension ---- */
void xr_pad_border(int w, int h, uint16_t *const planes[3],
                   const ptrdiff_t strides[3]) {
  /* whole 4:2:0 picture; border 80/40 as in yuv_pic.cc:39 */
  YuvPicture pic(ChromaFormat::k420, w, h, 10, true, 0, 0);
  FillPic(&pic, const_cast<const uint16_t *const *>(planes), strides);
  pic.PadBorder();
  ReadPic(pic, planes, strides, true);
}

/* ---- motion search ---- */
uint32_t xr_mvd_bits_fullpel(int mvp_x, int mvp_y, int fx, int fy, int down) {
  return InterSearch::GetMvdBitsFullpel(MotionVector(mvp_x, mvp_y), fx, fy,
                                        down);
}
uint32_t xr_mvd_bits(int mvp_x, int mvp_y, int mv_x, int mv_y, int down) {
  return InterSearch::GetMvdBits(MotionVector(mvp_x, mvp_y),
                                 MotionVector(mv_x, mv_y), down);
}

void xr_min_max_mv(int pos_x, int pos_y, int pic_w, int pic_h, int center_x,
                   int center_y, int search_range, int mv_min[2],
                   int mv_max[2]) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, 8);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, pos_x, pos_y, 8, 8);
  YuvPicture ref_pic(ChromaFormat::k420, pic_w, pic_h, 8, false, 0, 0);
  InterPrediction ip(Simd(8).inter_prediction, ref_pic, 8);
  MvFullpel mn, mx;
  ip.DetermineMinMaxMv(*cu, ref_pic, MotionVector(center_x, center_y),
                       search_range, &mn, &mx);
  mv_min[0] = mn.x; mv_min[1] = mn.y;
  mv_max[0] = mx.x; mv_max[1] = mx.y;
}

struct MeEnv {
  PictureData pic_data;
  YuvPicture orig_pic, ref_pic, rec_pic;
  EncoderSettings settings;
  ReferencePictureLists rpl;
  MeEnv(int bd, int pic_w, int pic_h, const uint16_t *orig, ptrdiff_t os,
        const uint16_t *ref, ptrdiff_t rs)
      : pic_data(ChromaFormat::k420, pic_w, pic_h, bd),
        orig_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0),
        ref_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0),
        rec_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0) {
    const uint16_t *po[3] = {orig, nullptr, nullptr};
    const uint16_t *pr[3] = {ref, nullptr, nullptr};
    ptrdiff_t so[3] = {os, 0, 0}, sr[3] = {rs, 0, 0};
    /* orig needs no border: copy the visible area only */
    for (int y = 0; y < pic_h; y++)
      std::memcpy(orig_pic.GetSamplePtr(YuvComponent::kY, 0, y), po[0] + y * so[0],
                  sizeof(Sample) * pic_w);
    FillPic(&ref_pic, pr, sr);
    settings.Initialize(SpeedMode::kSlow);
    pic_data.SetSubGopLength(16);
    pic_data.SetPoc(8);
  }
};

void xr_tz_search(int bd, const xvcgpu_me_block *b, int pic_w, int pic_h,
                  const uint16_t *orig, ptrdiff_t os, const uint16_t *ref,
                  ptrdiff_t rs, int out_mv[2], uint32_t *out_cost) {
  MeEnv env(bd, pic_w, pic_h, orig, os, ref, rs);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary,
                                         b->depth_nonzero ? 1 : 0, b->x, b->y,
                                         b->w, b->h);
  cu->SetFullpelMv(b->fullpel_mv != 0);
  InterPrediction ip(Simd(bd).inter_prediction, env.rec_pic, bd);
  /* a Qp whose sqrt(lambda) reproduces lambda16 exactly */
  double ls = (b->lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  /* make sure floor(65536*sqrt(lambda)) == lambda16 */
  assert(static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt())) ==
         b->lambda16);
  MotionVector mvp(b->mvp_x, b->mvp_y);
  MvFullpel mn, mx;
  ip.DetermineMinMaxMv(*cu, env.ref_pic, mvp, b->search_range, &mn, &mx);
  SampleMetric metric(Simd(bd).sample_metric, bd,
                      b->h > 8 ? MetricType::kSadFast : MetricType::kSad);
  TzSearch tz(env.orig_pic, ip, env.settings, b->search_range);
  MvFullpel r = tz.Search(*cu, qp, metric, mvp, env.ref_pic, mn, mx,
                          MvFullpel(b->prev_x, b->prev_y));
  out_mv[0] = r.x;
  out_mv[1] = r.y;
  if (out_cost) *out_cost = 0; /* not observable through the reference API */
}

void xr_subpel_search(int bd, const xvcgpu_me_block *b, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os, const uint16_t *ref,
                      ptrdiff_t rs, const int fullpel[2], int out_mv[2],
                      uint32_t *out_dist) {
  MeEnv env(bd, pic_w, pic_h, orig, os, ref, rs);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w,
                                         b->h);
  cu->SetFullpelMv(false);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl,
                 env.settings);
  double ls = (b->lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  SampleMetric metric(Simd(bd).sample_metric, bd, MetricType::kSatd);
  SampleBufferConst orig_buffer =
      env.orig_pic.GetSampleBuffer(YuvComponent::kY, b->x, b->y);
  SampleBufferStorage pred(64, 64);
  Distortion dist = 0;
  MotionVector mv = is.SubpelSearch(*cu, qp, metric, env.ref_pic,
                                    MotionVector(b->mvp_x, b->mvp_y),
                                    MvFullpel(fullpel[0], fullpel[1]),
                                    orig_buffer, &pred, &dist);
  out_mv[0] = mv.x;
  out_mv[1] = mv.y;
  if (out_dist) *out_dist = static_cast<uint32_t>(dist);
}

/* T6: InterSearch::EvalStartMvp / EvalFinalMvpIdx and the bit helpers */
int xr_eval_start_mvp(int bd, const xvcgpu_me_block *b, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os, const uint16_t *ref, ptrdiff_t rs,
                      const int32_t mvp[4], uint32_t *out_cost) {
  MeEnv env(bd, pic_w, pic_h, orig, os, ref, rs);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w, b->h);
  cu->SetFullpelMv(b->fullpel_mv != 0);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl, env.settings);
  double ls = (b->lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  cu->SetQp(qp);   /* CompareSample(cu, ...) reads the CU's qp (distortion weight) */
  std::array<MotionVector, constants::kNumInterMvPredictors> list;
  static_assert(constants::kNumInterMvPredictors == 2, "two predictors");
  list[0] = MotionVector(mvp[0], mvp[1]);
  list[1] = MotionVector(mvp[2], mvp[3]);
  SampleBufferStorage pred(64, 64);
  Distortion cost = 0;
  int idx = is.EvalStartMvp<false>(*cu, qp, list, env.ref_pic, &pred, &cost);
  if (out_cost) *out_cost = static_cast<uint32_t>(cost);
  return idx;
}

int xr_eval_final_mvp_idx(int fullpel_mv, const int32_t mvp[4], int mv_x, int mv_y, int start) {
  std::vector<uint16_t> dummy(64 * 64, 0);
  MeEnv env(10, 64, 64, dummy.data(), 64, dummy.data(), 64);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, 0, 0, 16, 16);
  cu->SetFullpelMv(fullpel_mv != 0);
  InterSearch is(Simd(10), env.pic_data, env.orig_pic, env.rec_pic, env.rpl, env.settings);
  std::array<MotionVector, constants::kNumInterMvPredictors> list;
  list[0] = MotionVector(mvp[0], mvp[1]);
  list[1] = MotionVector(mvp[2], mvp[3]);
  return is.EvalFinalMvpIdx(*cu, list, MotionVector(mv_x, mv_y), start);
}

/* T4: InterSearch::SearchMergeCandidates on a given candidate list (two
 * reference pictures: L0 index 0 and L1 index 0).  cands[5][5] = {inter_dir,
 * mv0_x, mv0_y, mv1_x, mv1_y}; returns the number to try, order in out_list. */
int xr_search_merge_candidates(int bd, int x, int y, int w, int h, int pic_w, int pic_h,
                               const uint16_t *orig, ptrdiff_t os, const uint16_t *ref0,
                               ptrdiff_t rs0, const uint16_t *ref1, ptrdiff_t rs1,
                               const int32_t *cands, double lambda_sqrt, int32_t *out_list,
                               uint64_t *out_dist);

/* T5: InterSearch::AffineGradientSearch / MotionEstAffine (uni-pred) */
void xr_affine_gradient_search(int bd, int width, int height, const uint16_t *pred,
                               ptrdiff_t ps, const int16_t *err, ptrdiff_t es, int mvd[4]) {
  std::vector<uint16_t> dummy(64 * 64, 0);
  MeEnv env(bd, 64, 64, dummy.data(), 64, dummy.data(), 64);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl,
                 env.settings);
  SampleBuffer pb(const_cast<uint16_t *>(pred), ps);
  ResidualBuffer eb(const_cast<int16_t *>(err), es);
  MvDelta2 r = is.AffineGradientSearch(width, height, pb, eb);
  mvd[0] = r[0].x;
  mvd[1] = r[0].y;
  mvd[2] = r[1].x;
  mvd[3] = r[1].y;
}

void xr_affine_me(int bd, const xvcgpu_affine_me_block *b, int pic_w, int pic_h,
                  const uint16_t *orig, ptrdiff_t os, const uint16_t *ref, ptrdiff_t rs,
                  const uint16_t *ref_other, ptrdiff_t ros, xvcgpu_affine_me_result *out) {
  MeEnv env(bd, pic_w, pic_h, orig, os, ref, rs);
  std::shared_ptr<const YuvPicture> refp(&env.ref_pic, [](const YuvPicture *) {});
  env.pic_data.GetRefPicLists()->SetRefPic(RefPicList::kL0, 0, 0, nullptr, refp, refp);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w, b->h);
  cu->SetUseAffine(true);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl,
                 env.settings);
  double ls = (b->lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  assert(static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt())) == b->lambda16);
  SampleBufferConst orig_buffer = env.orig_pic.GetSampleBuffer(YuvComponent::kY, b->x, b->y);
  SampleBufferStorage pred(64, 64);
  MotionVector3 mvp, boot, other;
  for (int i = 0; i < 3; i++) {
    mvp[i] = MotionVector(b->mvp[i][0], b->mvp[i][1]);
    boot[i] = MotionVector(b->bootstrap[i][0], b->bootstrap[i][1]);
    other[i] = MotionVector(b->other_mv[i][0], b->other_mv[i][1]);
  }
  const MotionVector3 *bootp = (b->flags & XVC_AFFINE_ME_HAS_BOOTSTRAP) ? &boot : nullptr;
  Distortion dist = 0;
  MotionVector3 mv;
  if (b->flags & XVC_AFFINE_ME_BIPRED) {
    /* SearchBiIterative :415-420: prediction of the other list, then
     * bipred_orig_buffer_.SubtractWeighted */
    YuvPicture other_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
    const uint16_t *pr[3] = {ref_other, nullptr, nullptr};
    ptrdiff_t sr[3] = {ros, 0, 0};
    FillPic(&other_pic, pr, sr);
    SampleBufferStorage other_pred(64, 64);
    is.MotionCompensationMv(*cu, YuvComponent::kY, other_pic, other, false, &other_pred);
    ResidualBufferStorage target(64, 64);
    target.SubtractWeighted(b->w, b->h, orig_buffer, other_pred);
    mv = is.MotionEstAffine(*cu, qp, InterSearch::SearchMethod::kTzSearch, RefPicList::kL0, 0,
                            true, target, mvp, bootp, &pred, &dist);
  } else {
    mv = is.MotionEstAffine(*cu, qp, InterSearch::SearchMethod::kTzSearch, RefPicList::kL0, 0,
                            false, orig_buffer, mvp, bootp, &pred, &dist);
  }
  for (int i = 0; i < 3; i++) {
    out->mv[i][0] = mv[i].x;
    out->mv[i][1] = mv[i].y;
  }
  out->dist = static_cast<uint32_t>(dist);
  out->iterations = 0; /* not observable through the reference API */
}

int xr_quant_fast2(int bd, int qp_raw, int intra_pic, int sign_hide, int scan_order,
                   int w, int h, const int16_t *in, ptrdiff_t is, int16_t *out,
                   ptrdiff_t os) {
  /* RdoQuant::QuantFast as shipped (sign hiding follows the restriction flag).
   * The scan order is a function of the CU (transform.cc:1614-1637): inter ->
   * diagonal; intra below 16x16 with a near-vertical mode -> horizontal scan,
   * near-horizontal mode -> vertical scan. */
  PictureData pic_data(ChromaFormat::k420, 64, 64, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, 0, 0, w, h);
  if (scan_order == 0) {
    cu->SetPredMode(PredictionMode::kInter);
  } else {
    cu->SetPredMode(PredictionMode::kIntra);
    cu->SetIntraModeLuma(IntraPrediction::Convert(
        scan_order == 1 ? IntraAngle::kVertical : IntraAngle::kHorizontal));
  }
  Qp qp = MakeQp(qp_raw, bd);
  EncoderSettings es;
  es.Initialize(SpeedMode::kSlow);
  RdoQuant rq(bd, es);
  Restrictions &r = Restrictions::GetRW();
  bool saved = r.disable_transform_sign_hiding;
  r.disable_transform_sign_hiding = !sign_hide;
  int nnz = rq.QuantFast(*cu, YuvComponent::kY, qp,
                         intra_pic ? PicturePredictionType::kIntra
                                   : PicturePredictionType::kBi,
                         in, is, out, os);
  r.disable_transform_sign_hiding = saved;
  return nnz;
}

void xr_qp_info(int qp_raw, int bd, int *qp_chroma, uint32_t *lambda16,
                double *chroma_dist_weight) {
  /* Y1: Qp as PictureData::Init builds it (picture_data.cc:91-106: lambda =
   * 0.57 * 2^((qp-12)/3), chroma table 1, offsets 0) and the full-pel search's
   * lambda16 = floor(65536 * sqrt(lambda)) (inter_tz_search.cc:98-99) */
  const double lambda = 0.57 * pow(2.0, (qp_raw - 12) / 3.0);
  Qp qp(qp_raw, ChromaFormat::k420, bd, lambda, 1, 0, 0);
  *qp_chroma = qp.GetQpRaw(YuvComponent::kU);
  *lambda16 = static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt()));
  *chroma_dist_weight = qp.GetDistortionWeight(YuvComponent::kU);
}

uint64_t xr_mc_metric(int bd, int metric_type, int qp_raw, int strength, int x, int y,
                      int w, int h,
                      int mv_x, int mv_y, int pic_w, int pic_h, const uint16_t *orig,
                      ptrdiff_t os, const uint16_t *ref, ptrdiff_t rs) {
  /* InterSearch::GetSubpelDist with an arbitrary metric */
  MeEnv env(bd, pic_w, pic_h, orig, os, ref, rs);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl,
                 env.settings);
  Qp qp = MakeQp(qp_raw, bd, 1.0);
  SampleMetric metric(Simd(bd).sample_metric, bd, MetricType(metric_type), strength);
  SampleBufferConst orig_buffer =
      env.orig_pic.GetSampleBuffer(YuvComponent::kY, x, y);
  SampleBufferStorage pred(64, 64);
  return is.GetSubpelDist(*cu, qp, env.ref_pic, metric, MotionVector(mv_x, mv_y),
                          orig_buffer, &pred);
}

void xr_full_search(int bd, int x, int y, int w, int h, int fullpel_mv,
                    int mvp_x, int mvp_y, uint32_t lambda16, const int mv_min[2],
                    const int mv_max[2], const int16_t *target, ptrdiff_t ts,
                    const uint16_t *ref, ptrdiff_t rs, int pic_w, int pic_h,
                    int out_mv[2]) {
  std::vector<uint16_t> dummy_orig(static_cast<size_t>(pic_w) * pic_h, 0);
  MeEnv env(bd, pic_w, pic_h, dummy_orig.data(), pic_w, ref, rs);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  cu->SetFullpelMv(fullpel_mv != 0);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, env.rpl,
                 env.settings);
  for (int yy = 0; yy < h; yy++)
    std::memcpy(is.bipred_orig_buffer_.GetDataPtr() +
                    yy * is.bipred_orig_buffer_.GetStride(),
                target + yy * ts, sizeof(int16_t) * w);
  double ls = (lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  SampleMetric metric(Simd(bd).sample_metric, bd,
                      h > 8 ? MetricType::kSadFast : MetricType::kSad);
  MvFullpel r = is.FullSearch(*cu, qp, metric, MotionVector(mvp_x, mvp_y),
                              env.ref_pic, MvFullpel(mv_min[0], mv_min[1]),
                              MvFullpel(mv_max[0], mv_max[1]));
  out_mv[0] = r.x;
  out_mv[1] = r.y;
}

/* ---- bi-prediction ---- */
struct BiEnv {
  PictureData pic_data;
  YuvPicture orig_pic, rec_pic;
  std::shared_ptr<YuvPicture> ref[2];
  std::shared_ptr<PictureData> ref_data;
  EncoderSettings settings;
  BiEnv(int bd, int pic_w, int pic_h)
      : pic_data(ChromaFormat::k420, pic_w, pic_h, bd),
        orig_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0),
        rec_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0) {
    for (int l = 0; l < 2; l++)
      ref[l] = std::make_shared<YuvPicture>(ChromaFormat::k420, pic_w, pic_h, bd,
                                            true, 0, 0);
    ref_data = std::make_shared<PictureData>(ChromaFormat::k420, 8, 8, bd);
    ref_data->SetNalType(NalUnitType::kPredictedPicture);
    settings.Initialize(SpeedMode::kSlow);
    pic_data.SetSubGopLength(16);
    pic_data.SetPoc(8);
    pic_data.SetNalType(NalUnitType::kBipredictedPicture);
    ReferencePictureLists *rpl = pic_data.GetRefPicLists();
    rpl->Reset(8);
    rpl->SetRefPic(RefPicList::kL0, 0, 0, ref_data, ref[0], nullptr);
    rpl->SetRefPic(RefPicList::kL1, 0, 16, ref_data, ref[1], nullptr);
  }
};

void xr_mc_bipred_block(int bd, int comp, int x, int y, int w, int h, int mv0_x,
                        int mv0_y, int mv1_x, int mv1_y, int pic_w, int pic_h,
                        const uint16_t *ref0, ptrdiff_t rs0, const uint16_t *ref1,
                        ptrdiff_t rs1, uint16_t *pred, ptrdiff_t ps) {
  BiEnv env(bd, pic_w, pic_h);
  const uint16_t *p0[3] = {nullptr, nullptr, nullptr}, *p1[3] = {nullptr, nullptr, nullptr};
  ptrdiff_t s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
  p0[comp] = ref0; s0[comp] = rs0;
  p1[comp] = ref1; s1[comp] = rs1;
  FillPic(env.ref[0].get(), p0, s0);
  FillPic(env.ref[1].get(), p1, s1);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  cu->SetPredMode(PredictionMode::kInter);
  cu->SetInterDir(InterDir::kBi);
  cu->SetRefIdx(0, RefPicList::kL0);
  cu->SetRefIdx(0, RefPicList::kL1);
  cu->SetMv(MotionVector(mv0_x, mv0_y), RefPicList::kL0);
  cu->SetMv(MotionVector(mv1_x, mv1_y), RefPicList::kL1);
  InterPrediction ip(Simd(bd).inter_prediction, env.rec_pic, bd);
  SampleBuffer pb(pred, ps);
  ip.MotionCompensation(*cu, YuvComponent(comp), &pb);
}

int xr_search_merge_candidates(int bd, int x, int y, int w, int h, int pic_w, int pic_h,
                               const uint16_t *orig, ptrdiff_t os, const uint16_t *ref0,
                               ptrdiff_t rs0, const uint16_t *ref1, ptrdiff_t rs1,
                               const int32_t *cands, double lambda_sqrt, int32_t *out_list,
                               uint64_t *out_dist) {
  BiEnv env(bd, pic_w, pic_h);
  for (int yy = 0; yy < pic_h; yy++)
    std::memcpy(env.orig_pic.GetSamplePtr(YuvComponent::kY, 0, yy), orig + yy * os,
                sizeof(Sample) * pic_w);
  const uint16_t *p0[3] = {ref0, nullptr, nullptr}, *p1[3] = {ref1, nullptr, nullptr};
  ptrdiff_t s0[3] = {rs0, 0, 0}, s1[3] = {rs1, 0, 0};
  FillPic(env.ref[0].get(), p0, s0);
  FillPic(env.ref[1].get(), p1, s1);
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  cu->SetPredMode(PredictionMode::kInter);
  Qp qp = MakeQp(32, bd, lambda_sqrt * lambda_sqrt);
  cu->SetQp(qp);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, *env.pic_data.GetRefPicLists(),
                 env.settings);
  TransformEncoder enc(Simd(bd), bd, 1, env.orig_pic, env.settings);
  BitWriter bw;
  SyntaxWriter writer(qp, PicturePredictionType::kBi, &bw);
  InterMergeCandidateList list;
  list.num = constants::kNumInterMergeCandidates;
  for (int m = 0; m < constants::kNumInterMergeCandidates; m++) {
    const int32_t *c = cands + 5 * m;
    list[m].inter_dir = c[0] == 2 ? InterDir::kBi : (c[0] == 1 ? InterDir::kL1 : InterDir::kL0);
    list[m].mv[0] = MotionVector(c[1], c[2]);
    list[m].mv[1] = MotionVector(c[3], c[4]);
    list[m].ref_idx[0] = 0;
    list[m].ref_idx[1] = 0;
  }
  InterSearch::MergeCandLookup lookup;
  lookup.fill(-1);
  const int num = is.SearchMergeCandidates(cu, qp, writer, list, &enc, &lookup);
  for (int m = 0; m < constants::kNumInterMergeCandidates; m++) out_list[m] = lookup[m];
  if (out_dist) {
    /* the per-candidate distortions of the same loop (:173-178), for diagnosis */
    SampleMetric metric(Simd(bd).sample_metric, bd, MetricType::kSatd);
    SampleBuffer pred_buffer = enc.GetPredBuffer(YuvComponent::kY);
    for (int m = 0; m < constants::kNumInterMergeCandidates; m++) {
      is.ApplyMergeCand(cu, list[m]);
      is.MotionCompensation(*cu, YuvComponent::kY, &pred_buffer);
      out_dist[m] = metric.CompareSample(*cu, YuvComponent::kY, env.orig_pic, pred_buffer);
    }
  }
  return num;
}

/* one step of SearchBiIterative on a prepared environment: searched list = L0
 * (ref[0]), other list = L1 (ref[1]) */
static void BipredStep(BiEnv *env, InterSearch *is, const xvcgpu_bi_block *j, int bd,
                       xvcgpu_me_result *out) {
  const xvcgpu_me_block *b = &j->blk;
  CodingUnit *cu = env->pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w, b->h);
  cu->SetPredMode(PredictionMode::kInter);
  cu->SetFullpelMv(b->fullpel_mv != 0);
  cu->SetRefIdx(0, RefPicList::kL0);
  cu->SetRefIdx(0, RefPicList::kL1);
  cu->SetMv(MotionVector(j->other_mv_x, j->other_mv_y), RefPicList::kL1);
  const YuvComponent comp = YuvComponent::kY;
  /* SearchBiIterative body, inter_search.cc:418-423 */
  cu->SetInterDir(InterDir::kL1);
  is->MotionCompensation(*cu, comp, &is->bipred_pred_buffer_);
  SampleBufferConst orig_luma = env->orig_pic.GetSampleBuffer(comp, b->x, b->y);
  is->bipred_orig_buffer_.SubtractWeighted(b->w, b->h, orig_luma, is->bipred_pred_buffer_);
  cu->SetInterDir(InterDir::kBi);
  double ls = (b->lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  MotionVector boot(j->boot_mv_x, j->boot_mv_y);
  SampleBufferStorage pred(64, 64);
  Distortion dist = 0;
  MotionVector mv = is->MotionEstNormal(
      *cu, qp, InterSearch::SearchMethod::kFullSearch, RefPicList::kL0, 0, true,
      is->bipred_orig_buffer_, MotionVector(b->mvp_x, b->mvp_y), &boot, &pred, &dist);
  out->fullpel_x = 0;
  out->fullpel_y = 0;
  out->mv_x = mv.x;
  out->mv_y = mv.y;
  out->fullpel_cost = 0;
  out->subpel_dist = static_cast<uint32_t>(dist);
  env->pic_data.ReleaseCu(cu);
}

/* n steps on the same three pictures (the environment is built once: what the
 * timing leg of bench.py uses) */
void xr_bipred_search_many(int bd, const xvcgpu_bi_block *jobs, int n, int pic_w, int pic_h,
                           const uint16_t *orig, ptrdiff_t os, const uint16_t *ref_other,
                           ptrdiff_t ros, const uint16_t *ref_search, ptrdiff_t rss,
                           xvcgpu_me_result *out) {
  BiEnv env(bd, pic_w, pic_h);
  for (int y = 0; y < pic_h; y++)
    std::memcpy(env.orig_pic.GetSamplePtr(YuvComponent::kY, 0, y), orig + y * os,
                sizeof(Sample) * pic_w);
  const uint16_t *ps[3] = {ref_search, nullptr, nullptr}, *po[3] = {ref_other, nullptr, nullptr};
  ptrdiff_t ss[3] = {rss, 0, 0}, so[3] = {ros, 0, 0};
  FillPic(env.ref[0].get(), ps, ss);
  FillPic(env.ref[1].get(), po, so);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic,
                 *env.pic_data.GetRefPicLists(), env.settings);
  for (int i = 0; i < n; i++) BipredStep(&env, &is, jobs + i, bd, out + i);
}

void xr_bipred_search(int bd, const xvcgpu_bi_block *j, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os, const uint16_t *ref_other,
                      ptrdiff_t ros, const uint16_t *ref_search, ptrdiff_t rss,
                      xvcgpu_me_result *out) {
  xr_bipred_search_many(bd, j, 1, pic_w, pic_h, orig, os, ref_other, ros, ref_search, rss, out);
}


/* InterSearch::SearchMotion (inter_search.cc:198-259) for one CU of a
 * bi-predicted picture with one reference per list: SearchRefIdx on L0 and L1,
 * SearchBiIterative (`iterations` refinement steps at most) and the final choice.
 * The closed-form bit costs (fast_inter_pred_bits, :1084-1130) are switched on so
 * that the result does not depend on entropy-coder state.  nb[2][2][2]: MVs of an
 * inter CU to the left (8 x h) and above (w x 8) [neighbour][list][x, y]; they
 * give the AMVP lists (inter_prediction.cc:139-248).  flags: 1 = full-pel MVs.
 * out[18]: cost, inter_dir (0 L0, 1 L1, 2 bi), mv0 x y, mv1 x y, mvp_idx0,
 * mvp_idx1, search_range L0, search_range L1, mvp list L0 {x0,y0,x1,y1}, L1;
 * out[18..25]: the uni-directional searches on their own (a second InterSearch):
 * cost, mv x y, mvp_idx of L0, then of L1. */
void xr_search_motion(int bd, int x, int y, int w, int h, int flags, uint32_t lambda16,
                      int iterations, int pic_w, int pic_h, const uint16_t *orig,
                      ptrdiff_t os, const uint16_t *ref0, ptrdiff_t rs0,
                      const uint16_t *ref1, ptrdiff_t rs1, const int32_t *nb,
                      int64_t *out) {
  BiEnv env(bd, pic_w, pic_h);
  env.settings.fast_inter_pred_bits = 1;
  env.settings.bipred_refinement_iterations = iterations;
  for (int yy = 0; yy < pic_h; yy++)
    std::memcpy(env.orig_pic.GetSamplePtr(YuvComponent::kY, 0, yy), orig + yy * os,
                sizeof(Sample) * pic_w);
  const uint16_t *p0[3] = {ref0, nullptr, nullptr}, *p1[3] = {ref1, nullptr, nullptr};
  ptrdiff_t s0[3] = {rs0, 0, 0}, s1[3] = {rs1, 0, 0};
  FillPic(env.ref[0].get(), p0, s0);
  FillPic(env.ref[1].get(), p1, s1);
  for (int k = 0; k < 2; k++) {
    if ((k == 0 && x < 8) || (k == 1 && y < 8)) continue;
    CodingUnit *n = k == 0 ? env.pic_data.CreateCu(CuTree::Primary, 1, x - 8, y, 8, h)
                           : env.pic_data.CreateCu(CuTree::Primary, 1, x, y - 8, w, 8);
    n->SetPredMode(PredictionMode::kInter);
    n->SetInterDir(InterDir::kBi);
    for (int l = 0; l < 2; l++) {
      n->SetRefIdx(0, static_cast<RefPicList>(l));
      n->SetMv(MotionVector(nb[4 * k + 2 * l], nb[4 * k + 2 * l + 1]),
               static_cast<RefPicList>(l));
    }
    env.pic_data.MarkUsedInPic(n);
  }
  CodingUnit *cu = env.pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  const double ls = (lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  cu->SetQp(qp);
  InterSearch is(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic, *env.pic_data.GetRefPicLists(),
                 env.settings);
  BitWriter bw;
  SyntaxWriter writer(qp, PicturePredictionType::kBi, &bw);
  SampleBufferStorage pred(64, 64);
  const InterSearchFlags sf =
      (flags & 1) ? InterSearchFlags::kFullPelMv : InterSearchFlags::kDefault;
  const Distortion cost = is.SearchMotion(cu, qp, writer, sf, &pred);
  out[0] = static_cast<int64_t>(cost);
  out[1] = cu->GetInterDir() == InterDir::kBi ? 2 : (cu->GetInterDir() == InterDir::kL1 ? 1 : 0);
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    const bool has = cu->HasMv(rl);
    out[2 + 2 * l] = has ? cu->GetMv(rl, MvCorner::kDefault).x : 0;
    out[3 + 2 * l] = has ? cu->GetMv(rl, MvCorner::kDefault).y : 0;
    out[6 + l] = has ? cu->GetMvpIdx(rl) : 0;
    out[8 + l] = is.GetSearchRangeUniPred(env.pic_data.GetRefPicLists()->GetRefPoc(rl, 0));
    const InterPredictorList mvp = is.GetMvpList(*cu, rl, 0);
    for (int k = 0; k < 2; k++) {
      out[10 + 4 * l + 2 * k] = mvp[k].x;
      out[11 + 4 * l + 2 * k] = mvp[k].y;
    }
  }
  InterSearch is2(Simd(bd), env.pic_data, env.orig_pic, env.rec_pic,
                  *env.pic_data.GetRefPicLists(), env.settings);
  SampleBufferConst orig_luma = env.orig_pic.GetSampleBuffer(YuvComponent::kY, x, y);
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    cu->ResetPredictionState();
    cu->SetPredMode(PredictionMode::kInter);
    cu->SetFullpelMv((flags & 1) != 0);
    cu->SetInterDir(l == 0 ? InterDir::kL0 : InterDir::kL1);
    CodingUnit::InterState st;
    const Distortion c = is2.SearchRefIdx(cu, qp, rl, writer, orig_luma,
                                          std::numeric_limits<Distortion>::max(), &pred, &st,
                                          nullptr);
    out[18 + 4 * l] = static_cast<int64_t>(c);
    out[19 + 4 * l] = cu->GetMv(rl, MvCorner::kDefault).x;
    out[20 + 4 * l] = cu->GetMv(rl, MvCorner::kDefault).y;
    out[21 + 4 * l] = cu->GetMvpIdx(rl);
  }
}


/* InterSearch::SearchMotion as the reference configures itself: up to three
 * reference pictures per list (default_num_ref_pics = 2, 3 in placebo;
 * encoder_settings.cc:36, :48), lists that may name the same pictures (list 1
 * then reuses list 0's search result, inter_search.cc:536-542), pictures with
 * only back references (PictureData::DetermineForceBipredL1MvdZero: list 1 of a
 * bi-directional CU carries no vector difference, :496-518, :410-413), the
 * closed-form bit prices (fast_inter_pred_bits, :1084-1130).
 *   pics: n_pics padded luma planes with their POCs; ref_pic[l][r] = index into
 *   pics of reference r of list l (num_ref[l] of them); cur_poc = the picture's POC.
 *   nb[2][2][3]: the inter CU to the left (8 x h) and above (w x 8):
 *   [neighbour][list]{ref_idx, mv_x, mv_y} (ref_idx < 0: list unused there).
 * out: [0] cost, [1] inter_dir, then per list {ref_idx, mv_x, mv_y, mvp_idx} at
 *   [2 + 4 l]; [10] force_l1_mvd_zero as the picture derives it; per (l, r) at
 *   [16 + 8 (3 l + r)]: search_range, the AMVP list {x0, y0, x1, y1}, same-POC index
 *   in list 0 (list 1 entries; -1 unique), 2 spare;
 *   [64 + 6 l]: the list's uni-directional search alone (a second InterSearch run
 *   L0 then L1, so list 1 sees list 0's results): cost, ref_idx, mv_x, mv_y,
 *   mvp_idx, cost of the best unique-POC reference (list 1). */
void xr_search_motion_multi(int bd, int x, int y, int w, int h, int flags, uint32_t lambda16,
                            int iterations, int pic_w, int pic_h, const uint16_t *orig,
                            ptrdiff_t os, int n_pics, const uint16_t *const *pics,
                            const ptrdiff_t *pic_strides, const int32_t *pic_pocs, int cur_poc,
                            const int32_t *num_ref, const int32_t *ref_pic, const int32_t *nb,
                            int64_t *out) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  YuvPicture orig_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  YuvPicture rec_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  std::vector<std::shared_ptr<YuvPicture>> store;
  for (int i = 0; i < n_pics; i++) {
    store.push_back(std::make_shared<YuvPicture>(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0));
    const uint16_t *p[3] = {pics[i], nullptr, nullptr};
    ptrdiff_t st[3] = {pic_strides[i], 0, 0};
    FillPic(store.back().get(), p, st);
  }
  auto ref_data = std::make_shared<PictureData>(ChromaFormat::k420, 8, 8, bd);
  ref_data->SetNalType(NalUnitType::kPredictedPicture);
  EncoderSettings settings;
  settings.Initialize(SpeedMode::kSlow);
  settings.fast_inter_pred_bits = 1;
  settings.bipred_refinement_iterations = iterations;
  pic_data.SetSubGopLength(16);
  pic_data.SetPoc(cur_poc);
  pic_data.SetNalType(NalUnitType::kBipredictedPicture);
  ReferencePictureLists *rpl = pic_data.GetRefPicLists();
  rpl->Reset(cur_poc);
  for (int l = 0; l < 2; l++)
    for (int r = 0; r < num_ref[l]; r++) {
      const int k = ref_pic[3 * l + r];
      rpl->SetRefPic(static_cast<RefPicList>(l), r, pic_pocs[k], ref_data, store[k], nullptr);
    }
  pic_data.force_bipred_l1_mvd_zero_ = pic_data.DetermineForceBipredL1MvdZero();
  for (int yy = 0; yy < pic_h; yy++)
    std::memcpy(orig_pic.GetSamplePtr(YuvComponent::kY, 0, yy), orig + yy * os,
                sizeof(Sample) * pic_w);
  for (int k = 0; k < 2; k++) {
    if ((k == 0 && x < 8) || (k == 1 && y < 8)) continue;
    const int32_t *q = nb + 6 * k;
    if (q[0] < 0 && q[3] < 0) continue;
    CodingUnit *n = k == 0 ? pic_data.CreateCu(CuTree::Primary, 1, x - 8, y, 8, h)
                           : pic_data.CreateCu(CuTree::Primary, 1, x, y - 8, w, 8);
    n->SetPredMode(PredictionMode::kInter);
    n->SetInterDir(q[0] >= 0 && q[3] >= 0 ? InterDir::kBi
                                          : (q[0] >= 0 ? InterDir::kL0 : InterDir::kL1));
    for (int l = 0; l < 2; l++) {
      if (q[3 * l] < 0) continue;
      n->SetRefIdx(q[3 * l], static_cast<RefPicList>(l));
      n->SetMv(MotionVector(q[3 * l + 1], q[3 * l + 2]), static_cast<RefPicList>(l));
    }
    pic_data.MarkUsedInPic(n);
  }
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
  const double ls = (lambda16 + 0.5) / 65536.0;
  Qp qp = MakeQp(32, bd, ls * ls);
  cu->SetQp(qp);
  InterSearch is(Simd(bd), pic_data, orig_pic, rec_pic, *rpl, settings);
  BitWriter bw;
  SyntaxWriter writer(qp, PicturePredictionType::kBi, &bw);
  SampleBufferStorage pred(64, 64);
  const InterSearchFlags sf =
      (flags & 1) ? InterSearchFlags::kFullPelMv : InterSearchFlags::kDefault;
  std::memset(out, 0, sizeof(int64_t) * 80);
  const Distortion cost = is.SearchMotion(cu, qp, writer, sf, &pred);
  out[0] = static_cast<int64_t>(cost);
  out[1] = cu->GetInterDir() == InterDir::kBi ? 2 : (cu->GetInterDir() == InterDir::kL1 ? 1 : 0);
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    const bool has = cu->HasMv(rl);
    out[2 + 4 * l] = has ? cu->GetRefIdx(rl) : -1;
    out[3 + 4 * l] = has ? cu->GetMv(rl, MvCorner::kDefault).x : 0;
    out[4 + 4 * l] = has ? cu->GetMv(rl, MvCorner::kDefault).y : 0;
    out[5 + 4 * l] = has ? cu->GetMvpIdx(rl) : 0;
  }
  out[10] = pic_data.GetForceBipredL1MvdZero();
  const std::vector<int> same = rpl->GetSamePocMappingFor(RefPicList::kL1);
  for (int l = 0; l < 2; l++)
    for (int r = 0; r < num_ref[l]; r++) {
      const RefPicList rl = static_cast<RefPicList>(l);
      int64_t *q = out + 16 + 8 * (3 * l + r);
      q[0] = is.GetSearchRangeUniPred(rpl->GetRefPoc(rl, r));
      cu->SetRefIdx(r, rl);
      const InterPredictorList mvp = is.GetMvpList(*cu, rl, r);
      for (int k = 0; k < 2; k++) {
        q[1 + 2 * k] = mvp[k].x;
        q[2 + 2 * k] = mvp[k].y;
      }
      q[5] = l == 1 ? same[r] : -1;
    }
  InterSearch is2(Simd(bd), pic_data, orig_pic, rec_pic, *rpl, settings);
  SampleBufferConst orig_luma = orig_pic.GetSampleBuffer(YuvComponent::kY, x, y);
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    if (l == 0) {
      cu->ResetPredictionState();
      cu->SetPredMode(PredictionMode::kInter);
      cu->SetFullpelMv((flags & 1) != 0);
    }
    cu->SetInterDir(l == 0 ? InterDir::kL0 : InterDir::kL1);
    CodingUnit::InterState st, st_unique;
    Distortion cost_unique = 0;
    const Distortion c = is2.SearchRefIdx(cu, qp, rl, writer, orig_luma,
                                          std::numeric_limits<Distortion>::max(), &pred, &st,
                                          l == 1 ? &st_unique : nullptr,
                                          l == 1 ? &cost_unique : nullptr);
    int64_t *q = out + 64 + 6 * l;
    q[0] = static_cast<int64_t>(c);
    q[1] = cu->GetRefIdx(rl);
    q[2] = cu->GetMv(rl, MvCorner::kDefault).x;
    q[3] = cu->GetMv(rl, MvCorner::kDefault).y;
    q[4] = cu->GetMvpIdx(rl);
    q[5] = l == 1 ? static_cast<int64_t>(cost_unique) : 0;
  }
}


/* ---- C1, decision half: InterSearch::CompressAndEvalCbf on a prepared CU -------
 * An environment (pictures, CU, motion, Qp, a SyntaxWriter in its picture-initial
 * state) that (a) runs the reference's own CompressAndEvalCbf (inter_search.cc:
 * 261-365) and reports every decision it made, and (b) answers the bit-price
 * questions the function asks its entropy coder, for a residual state the caller
 * describes (levels, cbf, transform choice) - the "bits" inputs of
 * xvcgpu_tx_eval_batch / xvcgpu_root_cbf_batch. */
struct XrC1Env {
  int bd, x, y, w, h;
  PictureData pic_data;
  YuvPicture orig_pic, rec_pic;
  std::shared_ptr<YuvPicture> ref[2];
  std::shared_ptr<PictureData> ref_data;
  EncoderSettings settings;
  Qp qp;
  int inter_dir, merge;
  MotionVector mv[2];
  XrC1Env(int bd_, int pic_w, int pic_h, int qp_raw, double lambda)
      : bd(bd_), pic_data(ChromaFormat::k420, pic_w, pic_h, bd_),
        orig_pic(ChromaFormat::k420, pic_w, pic_h, bd_, true, 0, 0),
        rec_pic(ChromaFormat::k420, pic_w, pic_h, bd_, true, 0, 0),
        qp(qp_raw, ChromaFormat::k420, bd_, lambda, 1, 0, 0) {
    for (int l = 0; l < 2; l++)
      ref[l] = std::make_shared<YuvPicture>(ChromaFormat::k420, pic_w, pic_h, bd_, true, 0, 0);
    ref_data = std::make_shared<PictureData>(ChromaFormat::k420, 8, 8, bd_);
    ref_data->SetNalType(NalUnitType::kPredictedPicture);
    settings.Initialize(SpeedMode::kSlow);
    pic_data.SetSubGopLength(16);
    pic_data.SetPoc(8);
    pic_data.SetNalType(NalUnitType::kBipredictedPicture);
    ReferencePictureLists *rpl = pic_data.GetRefPicLists();
    rpl->Reset(8);
    rpl->SetRefPic(RefPicList::kL0, 0, 0, ref_data, ref[0], nullptr);
    rpl->SetRefPic(RefPicList::kL1, 0, 16, ref_data, ref[1], nullptr);
  }
  CodingUnit *MakeCu() {
    CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, x, y, w, h);
    cu->SetQp(qp);
    cu->ResetPredictionState();
    cu->SetPredMode(PredictionMode::kInter);
    cu->SetMergeFlag(merge != 0);
    if (merge) cu->SetMergeIdx(0);
    cu->SetInterDir(inter_dir == 2 ? InterDir::kBi : (inter_dir == 1 ? InterDir::kL1 : InterDir::kL0));
    for (int l = 0; l < 2; l++) {
      const bool used = inter_dir == 2 || inter_dir == l;
      cu->SetRefIdx(used ? 0 : -1, static_cast<RefPicList>(l));
      cu->SetMv(used ? mv[l] : MotionVector(), static_cast<RefPicList>(l));
    }
    return cu;
  }
};

/* planes: orig, ref0, ref1 as [3] pointers at sample (0, 0) with strides (padded
 * references, >= 80 / 40 samples of border) */
void *xr_c1_create(int bd, int pic_w, int pic_h, int qp_raw, double lambda, int x, int y, int w,
                   int h, int inter_dir, int merge, const int32_t *mv,
                   const uint16_t *const *orig, const ptrdiff_t *orig_strides,
                   const uint16_t *const *ref0, const ptrdiff_t *ref0_strides,
                   const uint16_t *const *ref1, const ptrdiff_t *ref1_strides) {
  XrC1Env *e = new XrC1Env(bd, pic_w, pic_h, qp_raw, lambda);
  e->x = x; e->y = y; e->w = w; e->h = h;
  e->inter_dir = inter_dir;
  e->merge = merge;
  e->mv[0] = MotionVector(mv[0], mv[1]);
  e->mv[1] = MotionVector(mv[2], mv[3]);
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    for (int yy = 0; yy < e->orig_pic.GetHeight(comp); yy++)
      std::memcpy(e->orig_pic.GetSamplePtr(comp, 0, yy), orig[c] + yy * orig_strides[c],
                  sizeof(Sample) * e->orig_pic.GetWidth(comp));
  }
  FillPic(e->ref[0].get(), ref0, ref0_strides);
  FillPic(e->ref[1].get(), ref1, ref1_strides);
  return e;
}
void xr_c1_destroy(void *env) { delete static_cast<XrC1Env *>(env); }
/* What the device's jobs take from the Qp: out[0..2] raw qp per component,
 * [3..5] RdoQuant's lambda (fixed point), [6..8] its sign-hiding rd_factor
 * (rdo_quant.cc:251-252, :590-594); weights[0..2] Qp::GetDistortionWeight. */
void xr_c1_qp(void *env, int64_t *out, double *weights) {
  XrC1Env *e = static_cast<XrC1Env *>(env);
  for (int c = 0; c < 3; c++) {
    const YuvComponent yc = YuvComponent(c);
    out[c] = e->qp.GetQpRaw(yc);
    const double lam = e->qp.GetLambdaScaled(yc);
    const double inv_scale = e->qp.GetInvScale(yc);
    out[3 + c] = static_cast<int64_t>(lam * (1 << 16) + 0.5);
    out[6 + c] = static_cast<int64_t>(inv_scale * inv_scale / lam / 16 /
                                      (1ull << (2 * (e->bd - 8))) + 0.5);
    weights[c] = e->qp.GetDistortionWeight(yc);
  }
}

/* The reference's own CompressAndEvalCbf.  out: [0] returned distortion, [1..3]
 * cbf, [4] root_cbf, [5] transform select idx (-1 none), [6..8] transform skip,
 * [9] skip flag, [10..12] CRC-32 of the component's reconstruction block. */
void xr_c1_reference(void *env, uint64_t best_cu_cost, int fast_select, int64_t *out) {
  XrC1Env *e = static_cast<XrC1Env *>(env);
  e->settings.fast_transform_select_eval = fast_select;
  CodingUnit *cu = e->MakeCu();
  InterSearch is(Simd(e->bd), e->pic_data, e->orig_pic, e->rec_pic, *e->pic_data.GetRefPicLists(),
                 e->settings);
  TransformEncoder te(Simd(e->bd), e->bd, 3, e->orig_pic, e->settings);
  BitWriter bw;
  SyntaxWriter writer(e->qp, PicturePredictionType::kBi, &bw);
  const Distortion d = is.CompressAndEvalCbf(cu, e->qp, writer, best_cu_cost, &te, &e->rec_pic);
  out[0] = static_cast<int64_t>(d);
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    out[1 + c] = cu->GetCbf(comp);
    out[6 + c] = cu->GetTransformSkip(comp);
    uint32_t crc = 0;
    for (int yy = 0; yy < cu->GetHeight(comp); yy++)
      crc = xr_rd::Crc32(crc, e->rec_pic.GetSamplePtr(comp, cu->GetPosX(comp), cu->GetPosY(comp) + yy),
                         sizeof(Sample) * cu->GetWidth(comp));
    out[10 + c] = crc;
  }
  out[4] = cu->GetRootCbf();
  out[5] = cu->GetTransformSelectIdx();
  out[9] = cu->GetSkipFlag();
  e->pic_data.ReleaseCu(cu);
}

/* A residual state described by the caller -> the bits the reference's entropy
 * coder (picture-initial contexts, as in xr_c1_reference) prices it with.
 *   state: [0..2] cbf, [3..5] transform skip, [6] transform select idx (-1 none),
 *   [7] root cbf; levels[c]: the component's w_c x h_c levels, row-major (read
 *   when cbf[c]).
 *   kind 0: WriteResidualDataRdoCbf(comp)   (CompressAndEvalTransform, :84-88)
 *        1: WriteCbf(comp, false)            (:121-123)
 *        2: WriteRootCbf(false)              (inter_search.cc:270-275)
 *        3: GetCuBitsResidual                (transform_encoder.cc:287-296)
 *        4: GetCuBitsFull                    (:298-307) */
uint32_t xr_c1_bits(void *env, int kind, int comp, const int32_t *state,
                    const int16_t *const *levels) {
  XrC1Env *e = static_cast<XrC1Env *>(env);
  CodingUnit *cu = e->MakeCu();
  for (int c = 0; c < 3; c++) {
    const YuvComponent yc = YuvComponent(c);
    cu->SetCbf(yc, state[c] != 0);
    cu->SetTransformSkip(yc, state[3 + c] != 0);
    if (state[c] && levels && levels[c]) {
      CoeffBuffer dst = cu->GetCoeff(yc);
      const int cw = cu->GetWidth(yc), ch = cu->GetHeight(yc);
      for (int yy = 0; yy < ch; yy++)
        std::memcpy(dst.GetDataPtr() + yy * dst.GetStride(), levels[c] + yy * cw, sizeof(Coeff) * cw);
    }
  }
  cu->SetTransformFromSelectIdx(YuvComponent::kY, state[6]);
  cu->SetRootCbf(state[7] != 0);
  cu->SetSkipFlag(cu->GetMergeFlag() && !cu->GetHasAnyCbf());
  TransformEncoder te(Simd(e->bd), e->bd, 3, e->orig_pic, e->settings);
  CuWriter cu_writer(e->pic_data, nullptr);
  BitWriter bw;
  SyntaxWriter writer(e->qp, PicturePredictionType::kBi, &bw);
  RdoSyntaxWriter rdo(writer, 0);
  const YuvComponent yc = YuvComponent(comp);
  uint32_t bits = 0;
  switch (kind) {
    case 0: cu_writer.WriteResidualDataRdoCbf(*cu, yc, &rdo); bits = rdo.GetNumWrittenBits(); break;
    case 1: rdo.WriteCbf(*cu, yc, false); bits = rdo.GetNumWrittenBits(); break;
    case 2: rdo.WriteRootCbf(false); bits = rdo.GetNumWrittenBits(); break;
    case 3: bits = te.GetCuBitsResidual(*cu, writer, &cu_writer); break;
    case 4: bits = te.GetCuBitsFull(*cu, writer, &cu_writer); break;
  }
  e->pic_data.ReleaseCu(cu);
  return bits;
}


/* ---- the hot-path frame pass, executed by the reference's own classes ----
 * Same composition and argument block as xo_frame_pass (xvc_oracle_frame.c):
 * per CU TzSearch::Search + InterSearch::SubpelSearch +
 * InterPrediction::MotionCompensationMv for Y,U,V; per transform block
 * ResidualBuffer::Subtract -> ForwardTransform -> RdoQuant::QuantFast ->
 * Quantize::Inverse -> InverseTransform -> SampleBuffer::AddClip; then
 * DeblockingFilter::DeblockPicture, YuvPicture::PadBorder and
 * SampleMetric::ComparePicture.  Used to pin the oracle's whole composition
 * (tests/test_oracle_vs_ref.py) and as bench.py's cpu_baseline of kind
 * "reference".  me_results[].fullpel_cost stays 0 (not observable through the
 * reference API). */
static void ExtendBorder(uint16_t *p, ptrdiff_t st, int w, int h, int have,
                         int want) {
  /* the reference pads 80 / 40 samples; the argument block's planes carry
   * `want` >= that: continue the replication outwards */
  if (want <= have) return;
  for (int y = -have; y < h + have; y++) {
    uint16_t *row = p + y * st;
    for (int k = have + 1; k <= want; k++) {
      row[-k] = row[-have];
      row[w - 1 + k] = row[w - 1 + have];
    }
  }
  const size_t bytes = sizeof(uint16_t) * (w + 2 * want);
  for (int k = have + 1; k <= want; k++) {
    std::memcpy(p - k * st - want, p - have * st - want, bytes);
    std::memcpy(p + (h - 1 + k) * st - want, p + (h - 1 + have) * st - want,
                bytes);
  }
}

void XrLoadContexts(const xvcgpu_rdoq_contexts &c, Contexts *ctx);

void xr_frame_pass(xo_frame_args *a) {
  const int bd = a->bd, W = a->pic_w, H = a->pic_h;
  const int nthreads = a->threads > 1 ? a->threads : 1;
  (void)nthreads;
  YuvPicture orig_pic(ChromaFormat::k420, W, H, bd, true, 0, 0);
  YuvPicture ref_pic(ChromaFormat::k420, W, H, bd, true, 0, 0);
  YuvPicture rec_pic(ChromaFormat::k420, W, H, bd, true, 0, 0);
  for (int c = 0; c < 3; c++) {
    const int cs = c ? 1 : 0;
    for (int y = 0; y < (H >> cs); y++)
      std::memcpy(orig_pic.GetSamplePtr(YuvComponent(c), 0, y),
                  a->orig[c] + y * a->orig_stride[c], sizeof(Sample) * (W >> cs));
  }
  FillPic(&ref_pic, a->ref, a->ref_stride);
  EncoderSettings settings;
  settings.Initialize(SpeedMode::kSlow);
  const EncoderSimdFunctions &simd = Simd(bd); /* cached before the threads start */

  /* motion search + motion compensation: CUs are independent */
#pragma omp parallel num_threads(nthreads)
  {
    PictureData pd(ChromaFormat::k420, W, H, bd);
    pd.SetSubGopLength(16);
    pd.SetPoc(8);
    ReferencePictureLists rpl;
    InterPrediction ip(simd.inter_prediction, rec_pic, bd);
    InterSearch is(simd, pd, orig_pic, rec_pic, rpl, settings);
    SampleBufferStorage pred_tmp(64, 64);
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < a->n_cus; i++) {
      const xvcgpu_me_block *b = &a->me_blocks[i];
      xvcgpu_me_result *r = &a->me_results[i];
      CodingUnit *cu = pd.CreateCu(CuTree::Primary, b->depth_nonzero ? 1 : 0,
                                   b->x, b->y, b->w, b->h);
      cu->SetFullpelMv(b->fullpel_mv != 0);
      const double ls = (b->lambda16 + 0.5) / 65536.0;
      Qp qp = MakeQp(32, bd, ls * ls);
      const MotionVector mvp(b->mvp_x, b->mvp_y);
      MvFullpel mn, mx;
      ip.DetermineMinMaxMv(*cu, ref_pic, mvp, b->search_range, &mn, &mx);
      SampleMetric sad(simd.sample_metric, bd,
                       b->h > 8 ? MetricType::kSadFast : MetricType::kSad);
      TzSearch tz(orig_pic, ip, settings, b->search_range);
      const MvFullpel fp = tz.Search(*cu, qp, sad, mvp, ref_pic, mn, mx,
                                     MvFullpel(b->prev_x, b->prev_y));
      MotionVector mv(fp.x * 16, fp.y * 16);
      Distortion dist = 0;
      if (!b->fullpel_mv) {
        SampleMetric satd(simd.sample_metric, bd, MetricType::kSatd);
        SampleBufferConst orig_buffer =
            orig_pic.GetSampleBuffer(YuvComponent::kY, b->x, b->y);
        mv = is.SubpelSearch(*cu, qp, satd, ref_pic, mvp, fp, orig_buffer,
                             &pred_tmp, &dist);
      }
      r->fullpel_x = fp.x;
      r->fullpel_y = fp.y;
      r->fullpel_cost = 0;
      r->mv_x = mv.x;
      r->mv_y = mv.y;
      r->subpel_dist = static_cast<uint32_t>(dist);
      for (int c = 0; c < 3; c++) {
        const int cs = c ? 1 : 0;
        SampleBuffer pb(a->pred[c] + (b->y >> cs) * a->pred_stride[c] + (b->x >> cs),
                        a->pred_stride[c]);
        ip.MotionCompensationMv(*cu, YuvComponent(c), ref_pic, mv, false, &pb);
      }
      pd.ReleaseCu(cu);
    }
  }

  /* residual pipeline (transform_encoder.cc:203-285 with QuantFast) */
#pragma omp parallel num_threads(nthreads)
  {
    PictureData pd(ChromaFormat::k420, 64, 64, bd);
    ForwardTransform fwd(bd);
    InverseTransform inv(bd);
    Quantize quant;
    RdoQuant rq(bd, settings);
    std::vector<int16_t> resi(64 * 64), coeff(64 * 64), level(64 * 64), deq(64 * 64);
    Restrictions &restr = Restrictions::GetRW(); /* thread-local in the reference */
    const bool saved = restr.disable_transform_sign_hiding;
#pragma omp for schedule(dynamic, 48)
    for (int i = 0; i < a->n_tx; i++) {
      const xvcgpu_tx_block *t = &a->tx_blocks[i];
      const int c = t->comp, w = t->w, h = t->h;
      const int scan = (t->intra_pic >> XVC_TXF_SCAN_SHIFT) & 3;
      /* every block is run as the luma block of a CU of its own size with the
       * block's qp - exactly how the block-level functions were pinned */
      CodingUnit *cu = pd.CreateCu(CuTree::Primary, 0, 0, 0, w, h);
      if (scan == 0 && !t->dst4x4) {
        cu->SetPredMode(PredictionMode::kInter);
      } else {
        cu->SetPredMode(PredictionMode::kIntra);
        cu->SetIntraModeLuma(IntraPrediction::Convert(
            scan == 1 ? IntraAngle::kVertical
                      : (scan == 2 ? IntraAngle::kHorizontal : IntraAngle::kDc)));
      }
      const bool skip = t->tx_hor == XVC_TX_SKIP;
      if (!skip)
        cu->SetTransformType(YuvComponent::kY, static_cast<TransformType>(t->tx_ver),
                             static_cast<TransformType>(t->tx_hor));
      const ptrdiff_t os = a->orig_stride[c], ps = a->pred_stride[c], rs = a->rec_stride[c];
      SampleBufferConst ob(a->orig[c] + t->y * os + t->x, os);
      SampleBufferConst pb(a->pred[c] + t->y * ps + t->x, ps);
      SampleBuffer recb(a->rec[c] + t->y * rs + t->x, rs);
      ResidualBuffer rb(resi.data(), 64);
      CoeffBuffer cb(coeff.data(), 64);
      rb.Subtract(w, h, ob, pb);
      if (skip)
        fwd.TransformSkip(w, h, rb, &cb);
      else
        fwd.Transform(*cu, YuvComponent::kY, rb, &cb);
      Qp qp = MakeQp(t->qp, bd);
      restr.disable_transform_sign_hiding = (t->intra_pic & XVC_TXF_NO_SIGN_HIDING) != 0;
      int nnz;
      if (a->rdoq_params && (t->intra_pic & XVC_TXF_RDOQ)) {
        /* the quantiser the encoder really runs (transform_encoder.cc:230), on
         * the block's own component of a CU of the matching luma size, with the
         * Qp PictureData::Init would build from the picture qp and lambda */
        const xvcgpu_rdoq_params &prm = a->rdoq_params[i];
        const int cs = c ? 1 : 0;
        CodingUnit *rcu = pd.CreateCu(CuTree::Primary, 0, 0, 0, w << cs, h << cs);
        rcu->SetPredMode((prm.flags & XVC_RDOQ_INTRA_CU) ? PredictionMode::kIntra
                                                         : PredictionMode::kInter);
        Qp pic_qp(a->qp_y, ChromaFormat::k420, bd, a->rdoq_lambda, 1, 0, 0);
        const YuvComponent yc = YuvComponent(c);
        assert(pic_qp.GetQpRaw(yc) == t->qp);
        assert(static_cast<int64_t>(pic_qp.GetLambdaScaled(yc) * (1 << 16) + 0.5) == prm.lambda);
        {
          const double inv_scale = pic_qp.GetInvScale(yc);
          assert(static_cast<int64_t>(inv_scale * inv_scale / pic_qp.GetLambdaScaled(yc) / 16 /
                                      (1ull << (2 * (bd - 8))) + 0.5) == prm.rd_factor);
        }
        BitWriter bw;
        SyntaxWriter writer(pic_qp, PicturePredictionType::kUni, &bw);
        XrLoadContexts(a->rdoq_contexts[prm.ctx_index], &writer.ctx_);
        nnz = rq.QuantRdo(*rcu, yc, pic_qp, PicturePredictionType::kUni, writer, coeff.data(),
                          64, level.data(), w);
        pd.ReleaseCu(rcu);
      } else {
        nnz = rq.QuantFast(*cu, YuvComponent::kY, qp,
                           (t->intra_pic & XVC_TXF_INTRA_PIC) ? PicturePredictionType::kIntra
                                                              : PicturePredictionType::kBi,
                           coeff.data(), 64, level.data(), w);
      }
      a->nnz[i] = nnz;
      if (nnz) {
        quant.Inverse(YuvComponent::kY, qp, w, h, bd, level.data(), w, deq.data(), 64);
        CoeffBuffer db(deq.data(), 64);
        if (skip) {
          inv.TransformSkip(w, h, db, &rb);
        } else {
          cu->SetDcCoeffOnly(YuvComponent::kY, nnz == 1 && level[0] != 0);
          inv.Transform(*cu, YuvComponent::kY, db, &rb);
        }
        recb.AddClip(w, h, pb, rb, 0, static_cast<Sample>((1 << bd) - 1));
      } else {
        recb.CopyFrom(w, h, pb);
      }
      pd.ReleaseCu(cu);
    }
    restr.disable_transform_sign_hiding = saved;
  }

  /* CU metadata for the in-loop filter (plain bookkeeping, as in xo_frame_pass) */
  for (int i = 0; i < a->n_cus; i++) {
    const xvcgpu_me_block *b = &a->me_blocks[i];
    xvcgpu_cu_info *c = &a->cus[a->cu_base + i];
    std::memset(c, 0, sizeof(*c));
    c->x = static_cast<uint16_t>(b->x);
    c->y = static_cast<uint16_t>(b->y);
    c->w = b->w;
    c->h = b->h;
    c->cbf_luma = a->nnz[a->luma_tx_index ? a->luma_tx_index[i] : i] != 0;
    c->qp_y = static_cast<int8_t>(a->qp_y);
    c->qp_c = static_cast<int8_t>(a->qp_c);
    c->ref_poc[0] = a->ref_poc;
    c->ref_poc[1] = -1;
    for (int k = 0; k < 4; k++) {
      c->mv[0][k][0] = a->me_results[i].mv_x;
      c->mv[0][k][1] = a->me_results[i].mv_y;
    }
  }
  if (a->encode_only) return;

  /* in-loop filter over the whole picture's CU list, border, PSNR parts */
  int n_all = 0; /* the CU array covers the picture: count via the map */
  for (int y = 0; y < (H + 3) / 4; y++)
    for (int x = 0; x < (W + 3) / 4; x++)
      n_all = std::max(n_all, a->cu_map[y * a->map_stride + x] + 1);
  const int32_t l0[1] = {a->ref_poc};
  xr_deblock_picture(bd, W, H, 0, a->beta_offset, a->tc_offset, a->subblock, a->cus,
                     n_all, a->rec, a->rec_stride, l0, 1, nullptr, 0);
  FillPic(&rec_pic, a->rec, a->rec_stride);  /* visible area matters only */
  rec_pic.PadBorder();
  ReadPic(rec_pic, a->rec, a->rec_stride, true);
  for (int c = 0; c < 3; c++) {
    const int cs = c ? 1 : 0;
    const int have = static_cast<int>(
        (rec_pic.GetStride(YuvComponent(c)) - (W >> cs)) >> 1);
    ExtendBorder(a->rec[c], a->rec_stride[c], W >> cs, H >> cs, have, a->border[c]);
  }
  SampleMetric ssd(simd.sample_metric, bd, MetricType::kSsd);
  Qp pqp = MakeQp(32, bd);
  a->ssd[0] = ssd.ComparePicture(pqp, YuvComponent::kY, YuvComponent::kY, orig_pic,
                                 rec_pic);
  /* samples visited by the block walk: recovered from the reported PSNR */
  a->ssd[1] = 0;
  if (a->ssd[0]) {
    const double psnr =
        ssd.ComputePsnr(pqp, YuvComponent::kY, YuvComponent::kY, orig_pic, rec_pic);
    const double mse = 255.0 * 255.0 / std::pow(10.0, psnr / 10.0);
    a->ssd[1] = static_cast<uint64_t>(std::llround(a->ssd[0] / mse));
  }
}

/* ---- whole-picture passes around the hot path (oracle: xvc_oracle_stats.c) ---- */

/* Resampler::ConvertFrom: tightly packed planar 4:2:0 input (in_w x in_h at
 * in_bd) into an internal picture out_w x out_h at out_bd; returns the planes
 * without border, tightly packed uint16. */
void xr_import_picture(int in_bd, int out_bd, int in_w, int in_h, int out_w,
                       int out_h, const uint8_t *bytes, uint16_t *out_planes) {
  Resampler::SimdFunc rs;
  Resampler resampler(rs);
  PictureFormat fmt(in_w, in_h, in_bd, ChromaFormat::k420, ColorMatrix::kUndefined,
                    false);
  YuvPicture pic(ChromaFormat::k420, out_w, out_h, out_bd, true, out_w - in_w,
                 out_h - in_h);
  resampler.ConvertFrom(fmt, bytes, &pic);
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    for (int y = 0; y < pic.GetHeight(comp); y++) {
      std::memcpy(out_planes, pic.GetSamplePtr(comp, 0, y),
                  sizeof(Sample) * pic.GetWidth(comp));
      out_planes += pic.GetWidth(comp);
    }
  }
}

/* Resampler::ConvertTo without resizing: picture (w x h internal, display
 * size disp_w x disp_h) -> packed planar bytes at out_bd.  Returns bytes
 * written. */
size_t xr_export_picture(int bd, int out_bd, int dither, int w, int h, int disp_w,
                         int disp_h, const uint16_t *const planes[3],
                         const ptrdiff_t strides[3], uint8_t *out) {
  Resampler::SimdFunc rs;
  Resampler resampler(rs);
  YuvPicture pic(ChromaFormat::k420, w, h, bd, true, w - disp_w, h - disp_h);
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    for (int y = 0; y < pic.GetHeight(comp); y++)
      std::memcpy(pic.GetSamplePtr(comp, 0, y), planes[c] + y * strides[c],
                  sizeof(Sample) * pic.GetWidth(comp));
  }
  PictureFormat fmt(disp_w, disp_h, out_bd, ChromaFormat::k420,
                    ColorMatrix::kUndefined, dither != 0);
  std::vector<uint8_t> bytes;
  resampler.ConvertTo(pic, fmt, &bytes);
  std::memcpy(out, bytes.data(), bytes.size());
  return bytes.size();
}

/* Checksum(kCrc, mode).HashPicture */
int xr_picture_crc(int bd, int mode, int w, int h, const uint16_t *const planes[3],
                   const ptrdiff_t strides[3], uint8_t *hash) {
  YuvPicture pic(ChromaFormat::k420, w, h, bd, true, 0, 0);
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    for (int y = 0; y < pic.GetHeight(comp); y++)
      std::memcpy(pic.GetSamplePtr(comp, 0, y), planes[c] + y * strides[c],
                  sizeof(Sample) * pic.GetWidth(comp));
  }
  Checksum cs(Checksum::Method::kCrc, mode ? Checksum::Mode::kMaxRobust
                                           : Checksum::Mode::kMinOverhead);
  cs.HashPicture(pic);
  std::vector<uint8_t> hv = cs.GetHash();
  std::memcpy(hash, hv.data(), hv.size());
  return static_cast<int>(hv.size());
}

/* CuEncoder::CalcDeltaQpFromVariance for the CTU at (x, y); luma plane w x h
 * (multiples of 16: the reference reads past the plane otherwise). */
int xr_aqp_delta_qp(int bd, int w, int h, const uint16_t *luma, ptrdiff_t stride,
                    int x, int y, int ctu_size, int aqp_strength) {
  YuvPicture orig(ChromaFormat::k420, w, h, bd, false, 0, 0);
  YuvPicture rec(ChromaFormat::k420, w, h, bd, true, 0, 0);
  for (int yy = 0; yy < h; yy++)
    std::memcpy(orig.GetSamplePtr(YuvComponent::kY, 0, yy), luma + yy * stride,
                sizeof(Sample) * w);
  PictureData pic_data(ChromaFormat::k420, w, h, bd);
  EncoderSettings settings;
  settings.Initialize(SpeedMode::kSlow);
  settings.aqp_strength = aqp_strength;
  CuEncoder enc(Simd(bd), orig, &rec, &pic_data, settings);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, x, y, ctu_size, ctu_size);
  return enc.CalcDeltaQpFromVariance(cu);
}

/* PictureEncoder::DetermineAllowLic(kUni, one reference whose original is b) */
int xr_allow_lic(int bd, int w, int h, const uint16_t *a, ptrdiff_t sa,
                 const uint16_t *b, ptrdiff_t sb) {
  PictureFormat fmt(w, h, bd, ChromaFormat::k420, ColorMatrix::kUndefined, false);
  PictureEncoder enc(Simd(bd), fmt, 0, 0);
  auto ref_orig = std::make_shared<YuvPicture>(ChromaFormat::k420, w, h, bd, false, 0, 0);
  for (int y = 0; y < h; y++) {
    std::memcpy(enc.orig_pic_->GetSamplePtr(YuvComponent::kY, 0, y), a + y * sa,
                sizeof(Sample) * w);
    std::memcpy(ref_orig->GetSamplePtr(YuvComponent::kY, 0, y), b + y * sb,
                sizeof(Sample) * w);
  }
  ReferencePictureLists rpl;
  rpl.Reset(8);
  auto ref_data = std::make_shared<PictureData>(ChromaFormat::k420, 8, 8, bd);
  ref_data->SetNalType(NalUnitType::kPredictedPicture);
  rpl.SetRefPic(RefPicList::kL0, 0, 4, ref_data, nullptr, ref_orig);
  Restrictions &r = Restrictions::GetRW();
  const bool saved = r.disable_ext2_inter_local_illumination_comp;
  r.disable_ext2_inter_local_illumination_comp = false;
  const bool allow = enc.DetermineAllowLic(PicturePredictionType::kUni, rpl);
  r.disable_ext2_inter_local_illumination_comp = saved;
  return allow ? 1 : 0;
}

/* ---- intra prediction (oracle: xvc_oracle_intra.c) ---- */
static void IntraRefState(IntraPrediction *ip, int bd, const xvcgpu_intra_block *b,
                          const uint16_t *rec, ptrdiff_t rs,
                          IntraPrediction::RefState *state) {
  IntraPrediction::NeighborState nb;
  nb.has_above_left = (b->neighbors & XVC_INTRA_HAS_ABOVE_LEFT) != 0;
  nb.has_above = (b->neighbors & XVC_INTRA_HAS_ABOVE) != 0;
  nb.has_left = (b->neighbors & XVC_INTRA_HAS_LEFT) != 0;
  nb.has_above_right = b->above_right;
  nb.has_below_left = b->below_left;
  state->ref_samples.fill(0);
  state->ref_filtered.fill(0);
  ip->ComputeRefSamples(b->w, b->h, nb, rec + b->y * rs + b->x, rs,
                        &state->ref_samples[0], IntraPrediction::kRefSampleStride_);
  if (b->comp == 0)
    ip->FilterRefSamples(b->w, b->h, &state->ref_samples[0], &state->ref_filtered[0],
                         IntraPrediction::kRefSampleStride_);
  (void)bd;
}

/* IntraPrediction::ComputeRefSamples (+ FilterRefSamples for luma) + Predict for
 * one block; the neighbour availability comes from the job (the reference
 * derives it from its CU map in DetermineNeighbors). */
void xr_intra_pred_block(int bd, const xvcgpu_intra_block *b, int pic_w, int pic_h,
                         const uint16_t *rec, ptrdiff_t rs, uint16_t *pred, ptrdiff_t ps) {
  const int cs = b->comp ? 1 : 0;
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, b->x << cs, b->y << cs,
                                     b->w << cs, b->h << cs);
  YuvPicture rec_pic(ChromaFormat::k420, 8, 8, bd, false, 0, 0);
  IntraPrediction ip(bd);
  IntraPrediction::RefState state;
  IntraRefState(&ip, bd, b, rec, rs, &state);
  /* into a 64x64 scratch buffer, as the reference's callers do: horizontal
   * modes of non-square blocks are first written transposed, i.e. beyond the
   * block's h rows (intra_prediction.cc:447-451) */
  SampleBufferStorage tmp(64, 64);
  ip.Predict(static_cast<IntraMode>(b->mode), *cu, YuvComponent(b->comp), state, rec_pic,
             &tmp);
  for (int y = 0; y < b->h; y++)
    std::memcpy(pred + (b->y + y) * ps + b->x, tmp.GetDataPtr() + y * tmp.GetStride(),
                sizeof(Sample) * b->w);
}

/* The prediction + SATD loop of IntraSearch::DetermineSlowIntraModes for all 67
 * luma modes (no bits, no sorting). */
void xr_intra_satd_modes(int bd, const xvcgpu_intra_block *b, int pic_w, int pic_h,
                         const uint16_t *orig, ptrdiff_t os, const uint16_t *rec,
                         ptrdiff_t rs, uint32_t *dist) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w, b->h);
  YuvPicture rec_pic(ChromaFormat::k420, 8, 8, bd, false, 0, 0);
  YuvPicture orig_pic(ChromaFormat::k420, pic_w, pic_h, bd, false, 0, 0);
  for (int y = 0; y < pic_h; y++)
    std::memcpy(orig_pic.GetSamplePtr(YuvComponent::kY, 0, y), orig + y * os,
                sizeof(Sample) * pic_w);
  IntraPrediction ip(bd);
  IntraPrediction::RefState state;
  IntraRefState(&ip, bd, b, rec, rs, &state);
  SampleMetric satd(Simd(bd).sample_metric, bd, MetricType::kSatd);
  SampleBufferStorage pred(64, 64);
  Qp qp = MakeQp(32, bd);
  for (int m = 0; m < XVC_INTRA_NUM_MODES; m++) {
    ip.Predict(static_cast<IntraMode>(m), *cu, YuvComponent::kY, state, rec_pic, &pred);
    /* CompareSample(cu, comp, orig_pic, pred) minus the CU's Qp object (the
     * SATD does not use it) */
    dist[m] = static_cast<uint32_t>(satd.CompareSample(
        qp, YuvComponent::kY, b->w, b->h,
        orig_pic.GetSamplePtr(YuvComponent::kY, b->x, b->y),
        orig_pic.GetStride(YuvComponent::kY), pred.GetDataPtr(), pred.GetStride()));
  }
}

/* IntraPrediction::Predict(kLmChroma) for the U or V block of a CU: needs the
 * whole reconstruction picture (luma of the CU, neighbouring rows / columns). */
void xr_intra_lm_chroma(int bd, int comp, int x, int y, int w, int h, int pic_w, int pic_h,
                        const uint16_t *const planes[3], const ptrdiff_t strides[3],
                        uint16_t *out, ptrdiff_t os) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, 2 * x, 2 * y, 2 * w, 2 * h);
  YuvPicture rec_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  for (int c = 0; c < 3; c++) {
    const YuvComponent cc = YuvComponent(c);
    for (int yy = 0; yy < rec_pic.GetHeight(cc); yy++)
      std::memcpy(rec_pic.GetSamplePtr(cc, 0, yy), planes[c] + yy * strides[c],
                  sizeof(Sample) * rec_pic.GetWidth(cc));
  }
  IntraPrediction ip(bd);
  IntraPrediction::RefState state;   /* not used by the LM mode */
  SampleBufferStorage tmp(64, 64);
  /* the down-scaled luma is produced by the first chroma component's call and
   * reused by the second (intra_prediction.cc:574-580) */
  if (comp == 2)
    ip.Predict(IntraMode::kLmChroma, *cu, YuvComponent::kU, state, rec_pic, &tmp);
  ip.Predict(IntraMode::kLmChroma, *cu, YuvComponent(comp), state, rec_pic, &tmp);
  for (int yy = 0; yy < h; yy++)
    std::memcpy(out + yy * os, tmp.GetDataPtr() + yy * tmp.GetStride(), sizeof(Sample) * w);
}

/* InterPrediction::MotionCompensationMv(post_filter = true) of a CU with
 * use_lic: neighbouring CUs above / left are created in the CU map at the
 * positions the job names (their position enters through ClipMv). */
void xr_mc_lic_block(int bd, const xvcgpu_mc_lic_block *b, int above_w, int above_h,
                     int left_w, int left_h, int pic_w, int pic_h,
                     const uint16_t *const ref_planes[3], const ptrdiff_t ref_strides[3],
                     const uint16_t *const rec_planes[3], const ptrdiff_t rec_strides[3],
                     uint16_t *pred, ptrdiff_t ps) {
  PictureData pic_data(ChromaFormat::k420, pic_w, pic_h, bd);
  if (b->neighbors & XVC_LIC_HAS_ABOVE) {
    CodingUnit *a = pic_data.CreateCu(CuTree::Primary, 1, b->above_x, b->above_y,
                                      above_w, above_h);
    pic_data.MarkUsedInPic(a);
  }
  if (b->neighbors & XVC_LIC_HAS_LEFT) {
    CodingUnit *l = pic_data.CreateCu(CuTree::Primary, 1, b->left_x, b->left_y, left_w,
                                      left_h);
    pic_data.MarkUsedInPic(l);
  }
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 1, b->x, b->y, b->w, b->h);
  cu->SetUseLic(true);
  YuvPicture ref_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  YuvPicture rec_pic(ChromaFormat::k420, pic_w, pic_h, bd, true, 0, 0);
  FillPic(&ref_pic, ref_planes, ref_strides);
  for (int c = 0; c < 3; c++) {
    const YuvComponent cc = YuvComponent(c);
    for (int y = 0; y < rec_pic.GetHeight(cc); y++)
      std::memcpy(rec_pic.GetSamplePtr(cc, 0, y), rec_planes[c] + y * rec_strides[c],
                  sizeof(Sample) * rec_pic.GetWidth(cc));
  }
  InterPrediction ip(Simd(bd).inter_prediction, rec_pic, bd);
  const int cs = b->comp ? 1 : 0;
  SampleBuffer pb(pred + (b->y >> cs) * ps + (b->x >> cs), ps);
  Restrictions &r = Restrictions::GetRW();
  const bool saved = r.disable_ext2_inter_local_illumination_comp;
  r.disable_ext2_inter_local_illumination_comp = false;
  ip.MotionCompensationMv(*cu, YuvComponent(b->comp), ref_pic,
                          MotionVector(b->mv_x, b->mv_y), true, &pb);
  r.disable_ext2_inter_local_illumination_comp = saved;
}

/* ---- Q2: RdoQuant::QuantRdo (rdo_quant.cc:203-446) ------------------------- */
void XrLoadContexts(const xvcgpu_rdoq_contexts &c, Contexts *ctx);
namespace {
void LoadContexts(const xvcgpu_rdoq_contexts &c, Contexts *ctx) { XrLoadContexts(c, ctx); }
}  // namespace
void XrLoadContexts(const xvcgpu_rdoq_contexts &c, Contexts *ctx) {
  for (int i = 0; i < 2; i++) {
    ctx->coeff_ext.csbf_luma[i].state_ = c.csbf[0][i];
    ctx->coeff_ext.csbf_chroma[i].state_ = c.csbf[1][i];
  }
  for (int i = 0; i < 54; i++) ctx->coeff_ext.sig_luma[i].state_ = c.sig_luma[i];
  for (int i = 0; i < 12; i++) ctx->coeff_ext.sig_chroma[i].state_ = c.sig_chroma[i];
  for (int i = 0; i < 16; i++) ctx->coeff_ext.greater1_luma[i].state_ = c.greater1_luma[i];
  for (int i = 0; i < 6; i++) ctx->coeff_ext.greater1_chroma[i].state_ = c.greater1_chroma[i];
  for (int i = 0; i < 25; i++) {
    ctx->coeff_last_pos_x_luma[i].state_ = c.last_x_luma[i];
    ctx->coeff_last_pos_y_luma[i].state_ = c.last_y_luma[i];
  }
  for (int i = 0; i < 3; i++) {
    ctx->coeff_last_pos_x_chroma[i].state_ = c.last_x_chroma[i];
    ctx->coeff_last_pos_y_chroma[i].state_ = c.last_y_chroma[i];
  }
  ctx->cu_cbf_luma[0].state_ = c.cbf_luma;
  ctx->cu_cbf_chroma[0].state_ = c.cbf_chroma;
  ctx->cu_root_cbf[0].state_ = c.root_cbf;
}
namespace {
void StoreContexts(const Contexts &ctx, xvcgpu_rdoq_contexts *c) {
  std::memset(c, 0, sizeof(*c));
  for (int i = 0; i < 2; i++) {
    c->csbf[0][i] = ctx.coeff_ext.csbf_luma[i].state_;
    c->csbf[1][i] = ctx.coeff_ext.csbf_chroma[i].state_;
  }
  for (int i = 0; i < 54; i++) c->sig_luma[i] = ctx.coeff_ext.sig_luma[i].state_;
  for (int i = 0; i < 12; i++) c->sig_chroma[i] = ctx.coeff_ext.sig_chroma[i].state_;
  for (int i = 0; i < 16; i++) c->greater1_luma[i] = ctx.coeff_ext.greater1_luma[i].state_;
  for (int i = 0; i < 6; i++) c->greater1_chroma[i] = ctx.coeff_ext.greater1_chroma[i].state_;
  for (int i = 0; i < 25; i++) {
    c->last_x_luma[i] = ctx.coeff_last_pos_x_luma[i].state_;
    c->last_y_luma[i] = ctx.coeff_last_pos_y_luma[i].state_;
  }
  for (int i = 0; i < 3; i++) {
    c->last_x_chroma[i] = ctx.coeff_last_pos_x_chroma[i].state_;
    c->last_y_chroma[i] = ctx.coeff_last_pos_y_chroma[i].state_;
  }
  c->cbf_luma = ctx.cu_cbf_luma[0].state_;
  c->cbf_chroma = ctx.cu_cbf_chroma[0].state_;
  c->root_cbf = ctx.cu_root_cbf[0].state_;
}
}  // namespace

const uint32_t *xr_entropy_bits_table(void) { return &ContextModel::kEntropyBits_[0]; }

}  /* extern "C" (reopened below) */

namespace xr_seq {
uint32_t g_next = 0;
std::vector<uint32_t> g_of[9];
}  // namespace xr_seq

namespace xr_me {
bool g_capture = false;
int g_only_poc = -1;
std::set<int> g_also_poc;
std::vector<Call> g_calls;
}  // namespace xr_me

namespace xvc {
MvFullpel ObservedTzSearch::Search(const CodingUnit &cu, const Qp &qp, const SampleMetric &metric,
                                   const MotionVector &mvp, const YuvPicture &ref_pic,
                                   const MvFullpel &mv_min, const MvFullpel &mv_max,
                                   const MvFullpel &prev_search) {
  const MvFullpel best = real_.Search(cu, qp, metric, mvp, ref_pic, mv_min, mv_max, prev_search);
  const int poc = static_cast<int>(cu.GetPicData()->GetPoc());
  if (!xr_me::g_capture || (xr_me::g_only_poc >= 0 && poc != xr_me::g_only_poc &&
                            !xr_me::g_also_poc.count(poc)))
    return best;
  const YuvComponent comp = YuvComponent::kY;
  xr_me::Call c;
  std::memset(&c, 0, sizeof(c));
  c.poc = poc;
  c.ref_poc = -1;
  const ReferencePictureLists *rpl = cu.GetRefPicLists();
  for (int l = 0; l < 2 && c.ref_poc < 0; l++) {
    const RefPicList list = l ? RefPicList::kL1 : RefPicList::kL0;
    for (int i = 0; i < rpl->GetNumRefPics(list); i++)
      if (rpl->GetRefPic(list, i) == &ref_pic) {
        c.ref_poc = static_cast<int>(rpl->GetRefPoc(list, i));
        break;
      }
  }
  c.x = static_cast<int16_t>(cu.GetPosX(comp));
  c.y = static_cast<int16_t>(cu.GetPosY(comp));
  c.w = static_cast<uint8_t>(cu.GetWidth(comp));
  c.h = static_cast<uint8_t>(cu.GetHeight(comp));
  c.depth_nonzero = cu.GetDepth() != 0;
  c.fullpel_mv = cu.GetFullpelMv();
  c.use_lic = cu.GetUseLic() ? 1 : 0;
  c.mvp_x = mvp.x;
  c.mvp_y = mvp.y;
  c.prev_x = prev_search.x;
  c.prev_y = prev_search.y;
  c.lambda16 = static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt()));
  c.search_range = search_range_;
  c.fullpel_x = best.x;
  c.fullpel_y = best.y;
  /* the sub-pel stage from this full-pel position, by the reference's own code */
  InterSearch &is = const_cast<InterSearch &>(static_cast<const InterSearch &>(inter_pred_));
  SampleMetric subpel_metric(is.simd_.sample_metric, is.bitdepth_, is.GetSubpelMetric(cu));
  auto orig_buffer = orig_pic_.GetSampleBuffer(comp, cu.GetPosX(comp), cu.GetPosY(comp));
  SampleBufferStorage pred_storage(constants::kMaxBlockSize, constants::kMaxBlockSize);
  SampleBuffer &pred = pred_storage;
  Distortion dist = 0;
  MotionVector mv;
  if (cu.GetFullpelMv()) {
    mv = MotionVector(best);
    dist = is.GetSubpelDist(cu, qp, ref_pic, subpel_metric, mv, orig_buffer, &pred);
  } else {
    mv = is.SubpelSearch(cu, qp, subpel_metric, ref_pic, mvp, best, orig_buffer, &pred, &dist);
  }
  c.mv_x = mv.x;
  c.mv_y = mv.y;
  c.dist = static_cast<uint32_t>(dist);
  xr_me::g_calls.push_back(c);
  xr_seq::Stamp(0);
  return best;
}
}  // namespace xvc

extern "C" {

/* Capture control for the hook above: start (only_poc < 0: every picture),
 * then run an encode (xr_stream_encode), then read the records. */
void xr_me_capture_begin(int only_poc) {
  xr_me::g_calls.clear();
  xr_seq::g_of[0].clear();
  xr_me::g_also_poc.clear();
  xr_me::g_only_poc = only_poc;
  xr_me::g_capture = true;
}
void xr_me_capture_also(int poc) { xr_me::g_also_poc.insert(poc); }
long xr_me_capture_end(void) {
  xr_me::g_capture = false;
  return static_cast<long>(xr_me::g_calls.size());
}
int xr_me_call_size(void) { return static_cast<int>(sizeof(xr_me::Call)); }
const void *xr_me_calls(void) { return xr_me::g_calls.data(); }

/* Sub-GOP arithmetic (segment_header.cc:135-175) as the encoder calls it:
 * sub_gop_start_poc = the POC the picture's sub-GOP starts after (encoder.cc:97). */
static int XrSubGopStart(int n, int len) { return n < 1 ? 0 : (n - 1) / len * len; }
int xr_doc_from_poc(int poc, int sub_gop_length) {
  return static_cast<int>(
      SegmentHeader::CalcDocFromPoc(poc, sub_gop_length, XrSubGopStart(poc, sub_gop_length)));
}
int xr_poc_from_doc(int doc, int sub_gop_length) {
  return static_cast<int>(
      SegmentHeader::CalcPocFromDoc(doc, sub_gop_length, XrSubGopStart(doc, sub_gop_length)));
}
int xr_tid_from_doc(int doc, int sub_gop_length) {
  return SegmentHeader::CalcTidFromDoc(doc, sub_gop_length, XrSubGopStart(doc, sub_gop_length));
}

/* The context states a fresh SyntaxWriter starts a picture with
 * (CabacContexts::ResetStates for the picture qp / type). */
void xr_rdoq_init_contexts(int bd, int qp_raw, int pic_type, xvcgpu_rdoq_contexts *out) {
  Qp qp = MakeQp(qp_raw, bd);
  BitWriter bw;
  SyntaxWriter writer(qp, static_cast<PicturePredictionType>(pic_type), &bw);
  StoreContexts(writer.GetContexts(), out);
}

/* RdoQuant::QuantRdo for component `comp` of a CU whose component block is
 * w x h, with the given context states.  qp_raw_luma / lambda build the Qp as
 * PictureData::Init does; the component's raw qp and the two host-side
 * constants of xvcgpu_rdoq_params (as the reference's doubles give them) are
 * returned so that the oracle can be fed the same inputs.  prm_io->flags in;
 * lambda / rd_factor out. */
int xr_quant_rdo(int bd, int qp_raw_luma, double lambda, int comp, int scan_order, int sign_hide,
                 int w, int h, const xvcgpu_rdoq_contexts *ctx, xvcgpu_rdoq_params *prm_io,
                 int *comp_qp_raw, const int16_t *src, ptrdiff_t is, int16_t *out,
                 ptrdiff_t os) {
  PictureData pic_data(ChromaFormat::k420, 64, 64, bd);
  const int s = comp ? 1 : 0;
  CodingUnit *cu = pic_data.CreateCu(CuTree::Primary, 0, 0, 0, w << s, h << s);
  const bool intra = (prm_io->flags & XVC_RDOQ_INTRA_CU) != 0;
  cu->SetPredMode(intra ? PredictionMode::kIntra : PredictionMode::kInter);
  if (intra) {
    cu->SetIntraModeLuma(scan_order == 1   ? IntraPrediction::Convert(IntraAngle::kVertical)
                         : scan_order == 2 ? IntraPrediction::Convert(IntraAngle::kHorizontal)
                                           : IntraMode::kDc);
    cu->SetIntraModeChroma(IntraChromaMode::kDmChroma);
  }
  Qp qp(qp_raw_luma, ChromaFormat::k420, bd, lambda, 1, 0, 0);
  const YuvComponent yc = YuvComponent(comp);
  *comp_qp_raw = qp.GetQpRaw(yc);
  prm_io->lambda = static_cast<int64_t>(qp.GetLambdaScaled(yc) * (1 << 16) + 0.5);
  {
    const double lam = qp.GetLambdaScaled(yc);
    const double inv_scale = qp.GetInvScale(yc);
    prm_io->rd_factor = static_cast<int64_t>(inv_scale * inv_scale / lam / 16 /
                                             (1ull << (2 * (bd - 8))) + 0.5);
  }
  BitWriter bw;
  SyntaxWriter writer(qp, PicturePredictionType::kBi, &bw);
  LoadContexts(*ctx, &writer.ctx_);
  EncoderSettings es;
  es.Initialize(SpeedMode::kSlow);
  es.rdo_quant_2x2 = (prm_io->flags & XVC_RDOQ_NO_2X2) ? 0 : 1;
  RdoQuant rq(bd, es);
  Restrictions &r = Restrictions::GetRW();
  const bool saved = r.disable_transform_sign_hiding;
  r.disable_transform_sign_hiding = !sign_hide;
  assert(static_cast<int>(TransformHelper::DetermineScanOrder(*cu, yc)) == scan_order);
  const int nnz = rq.QuantRdo(*cu, yc, qp, PicturePredictionType::kBi, writer, src, is, out, os);
  r.disable_transform_sign_hiding = saved;
  return nnz;
}
}  // extern "C"

/* ---- RD-search capture (tools/gen_rd_golden.py) ----------------------------
 * Records, while the reference encoder codes a picture, every bi-prediction
 * refinement step, affine motion search, merge-candidate ranking and inter-CU
 * TransformAndReconstruct it makes - inputs and results - through the hooks
 * declared at the top of this file.  Layouts mirrored by numpy dtypes in
 * tools/gen_rd_golden.py (sizes checked). */
namespace xr_rd {
using namespace xvc;  // NOLINT

enum { kBi = 1, kAffineUni = 2, kAffineBi = 3 };
enum { kFlagFullpel = 1, kFlagLic = 2, kFlagHasBoot = 4, kFlagAffine = 8, kFlagMerge = 16,
       kFlagSkip = 32 };

struct MeStep {            /* one MotionEstimation call of SearchRefIdx */
  int32_t poc;
  int16_t x, y;
  uint8_t w, h, kind, flags;
  uint8_t list;            /* the searched list */
  int8_t ref_idx, other_ref_idx;
  uint8_t start_mvp_idx, final_mvp_idx, pad[3];
  int32_t ref_poc, other_ref_poc;
  uint32_t lambda16;
  int32_t mvp[2][3][2];    /* the list's two predictors ([k][0] for a plain vector) */
  int32_t boot[3][2];
  int32_t other_mv[3][2];  /* the other list's vector(s) the target was formed with */
  int32_t mv[3][2];        /* result */
  uint32_t dist;           /* *out_dist */
  int32_t nb_index;        /* LIC CUs: which Neighbours record (-1: none) */
};

struct MergeCall {         /* one SearchMergeCandidates */
  int32_t poc;
  int16_t x, y;
  uint8_t w, h, pad[2];
  double lambda_sqrt;
  uint8_t inter_dir[5], use_lic[5];
  int8_t ref_idx[5][2];
  int32_t ref_poc[5][2];
  int32_t mv[5][2][2];
  int32_t order[5];        /* candidate indices after the stable sort */
  double cost[5];          /* their costs, same order */
  int32_t num;             /* the function's return value (:186-195) */
  int32_t nb_index;        /* a candidate with use_lic: which Neighbours record (else -1) */
};

struct Eval {              /* the CU state a group of transform calls shares */
  int32_t poc;
  int16_t x, y;
  uint8_t w, h, inter_dir, flags;
  int8_t ref_idx[2];
  int8_t qp[3];            /* qp.GetQpRaw(comp) */
  uint8_t pad;
  int32_t ref_poc[2];
  int32_t mv[2][3][2];
  int32_t ctx_index;       /* which context snapshot */
  int32_t qp_index;        /* which QpParams */
  int32_t nb_index;        /* LIC CUs: which Neighbours record (-1: none) */
  int32_t pad2;
  uint64_t dist_zero[3];   /* CompressAndEvalTransform's cbf-zero distortion (~0: not evaluated) */
};

struct QpParams {          /* what the quantiser / metrics take from the Qp */
  int8_t qp_raw[3];
  uint8_t pad[5];
  int64_t lambda[3];       /* (int64)(GetLambdaScaled(comp) * 65536 + 0.5) */
  int64_t rd_factor[3];    /* rdo_quant.cc:590-594 */
  double dist_weight[3];   /* Qp::GetDistortionWeight(comp) */
};

/* What the local illumination model of a CU reads besides the reference
 * picture (DeriveLicParams, inter_prediction.cc:1577-1663): the CUs above / left
 * (their positions: ClipMv) and the CURRENT reconstruction's row above / column
 * left of the block, per component - state of the RD search at that moment, not
 * of the final picture.  samples: for comp 0,1,2: the row (w_c samples, if
 * has_above) then the column (h_c samples, if has_left), at sample_off in the
 * sample array. */
struct Neighbours {
  int16_t x, y;
  uint8_t w, h, has_above, has_left;
  int16_t above_x, above_y, left_x, left_y;
  uint32_t sample_off, sample_count;
};

struct TxCall {            /* one TransformAndReconstruct */
  int32_t eval;
  uint8_t comp, tx_skip, tx_hor, tx_ver, scan, completed;
  int8_t tx_select_idx;
  uint8_t pad;
  int32_t nnz;             /* QuantRdo's return value */
  uint32_t levels_crc;     /* CRC-32 of the w x h levels, row-major */
  uint32_t rec_crc;        /* CRC-32 of the reconstruction block (completed calls) */
  uint32_t pad2;
  uint64_t dist;           /* the returned distortion (completed calls) */
};

/* round 4: every candidate SearchRefIdx prices (inter_search.cc:556-571) - uni
 * searches, list 1's re-use of list 0's result, bi-prediction steps, affine - with
 * the bits GetInterPredBits returned for it through the throw-away RdoSyntaxWriter
 * (:1131-1135, the default: fast_inter_pred_bits == 0) and the context states
 * CuWriter::WriteInterPrediction (cu_writer.cc:122-172) read for this CU. */
struct Cand {
  int32_t poc;
  int16_t x, y;
  uint8_t w, h, kind, flags;   /* kind: 0 uni, kBi, kAffineUni, kAffineBi; flags as MeStep */
  uint8_t list;
  int8_t ref_idx;
  uint8_t reused;              /* list 1 took list 0's result (:536-542) */
  uint8_t mvp_idx;             /* after EvalFinalMvpIdx */
  uint8_t inter_dir;           /* cu->GetInterDir() while the bits were written */
  int8_t other_ref_idx;        /* bi: the other list's picture */
  uint8_t other_mvp_idx, force_mvd_zero_other;
  int32_t mv[3][2];
  int32_t mvp[2][3][2];        /* the list's two predictors (GetMvpList; [k][0] plain) */
  uint8_t start_mvp_idx;       /* uni: EvalStartMvp's choice; bi: the uni search's final one */
  uint8_t pad[3];
  int32_t other_mvd[2][2];     /* bi: the other list's mvd(s) (affine: two corners) */
  uint32_t dist;
  uint32_t bits;               /* GetInterPredBits(*cu, bitstream_writer) */
  uint32_t lambda16;
  int32_t ictx_index;          /* which InterContexts snapshot */
};
struct Final {                 /* the CU's motion state as SearchMotion returns it */
  int32_t poc;
  int16_t x, y;
  uint8_t w, h, which, flags;  /* which: 0 bi, 1 list 0, 2 list 1 (unique picture) */
  uint8_t inter_dir;
  int8_t ref_idx[2];
  uint8_t mvp_idx[2];
  uint8_t pad[3];
  int32_t mv[2][3][2];
  int32_t mvd[2][2][2];
};
std::vector<Cand> g_cands;
std::vector<Final> g_finals;
std::vector<xvcgpu_inter_contexts> g_ictx;
std::unordered_map<std::string, int> g_ictx_index;

bool g_capture = false;
int g_only_poc = -1;
std::vector<MeStep> g_steps;
std::vector<MergeCall> g_merges;
std::vector<Eval> g_evals;
std::vector<QpParams> g_qps;
std::vector<TxCall> g_calls;
std::vector<xvcgpu_rdoq_contexts> g_ctx;
std::vector<Neighbours> g_nb;
std::vector<uint16_t> g_nb_samples;
std::unordered_map<std::string, int> g_ctx_index, g_qp_index, g_nb_index;
long g_skipped_intra = 0;
const CodingUnit *g_pending_cu = nullptr;
int g_pending_call = -1;

/* TransformAndReconstruct of INTRA CUs, a sample (every g_itx_stride-th call; LM
 * chroma left out: its prediction reads the co-located luma block): the block,
 * its prediction mode, what DetermineNeighbors sees and the reconstruction's
 * reference samples around it - the prediction is the caller's, but nothing
 * outside the CU changes between FillReferenceState and this call -, the
 * transform choice, the context snapshot, and the results. */
struct IntraTx {
  int32_t poc;
  int16_t x, y;              /* in samples of comp */
  uint8_t w, h, comp, mode;  /* mode: IntraMode of the component (0..66) */
  uint8_t neighbors, above_right, below_left, tx_skip;
  uint8_t tx_hor, tx_ver, scan, dst4x4;
  uint8_t completed, intra_pic;
  int8_t qp, qp_luma;        /* raw qp of the component / of luma (structural SSD) */
  int32_t ctx_index, qp_index;
  int32_t nnz;
  uint32_t levels_crc, pred_crc, rec_crc;
  int32_t sample_off;        /* [above-left] [above: w + above_right] [left: h + below_left] */
  uint64_t dist;
};
std::vector<IntraTx> g_itx;
std::vector<uint16_t> g_itx_samples;
int g_itx_stride = 0, g_itx_cap = 0;
long g_itx_seen = 0;
int g_pending_itx = -1;
/* round 5 (tools/gen_order_golden.py): EVERY such call of the wanted picture, in the global
 * order (xr_seq table 8), LM chroma included - after its reference samples the luma
 * rectangle RescaleLuma reads (intra_prediction.cc:873-906: rows y - 2 .. y + h - 1 when
 * there is a row above, columns x - 3 .. x + w - 1 when there is a column left; luma
 * units) -, with context / QP tables of their own so that the inter tables' indices do
 * not move */
bool g_itx_walk = false;
int g_itx_poc = -1;
std::vector<xvcgpu_rdoq_contexts> g_itx_ctx;
std::vector<QpParams> g_itx_qps;
std::unordered_map<std::string, int> g_itx_ctx_index, g_itx_qp_index;

static bool Wanted(const CodingUnit &cu) {
  return g_capture &&
         (g_only_poc < 0 || static_cast<int>(cu.GetPicData()->GetPoc()) == g_only_poc);
}

uint32_t Crc32(uint32_t crc, const void *data, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  const uint8_t *p = static_cast<const uint8_t *>(data);
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
  return ~crc;
}

static void PutMv(int32_t out[3][2], const MotionVector &m) {
  out[0][0] = m.x;
  out[0][1] = m.y;
}
static void PutMv(int32_t out[3][2], const MotionVector3 &m) {
  for (int k = 0; k < 3; k++) {
    out[k][0] = m[k].x;
    out[k][1] = m[k].y;
  }
}
static void OtherMv(const CodingUnit &cu, RefPicList l, const MotionVector *, int32_t out[3][2]) {
  PutMv(out, cu.GetMv(l, MvCorner::kDefault));
}
static void OtherMv(const CodingUnit &cu, RefPicList l, const MotionVector3 *, int32_t out[3][2]) {
  PutMv(out, cu.GetMvAffine(l));
}

static int NeighbourIndex(const CodingUnit &cu, const YuvPicture &rec_pic, bool force = false) {
  if (!cu.GetUseLic() && !force) return -1;
  const YuvComponent luma = YuvComponent::kY;
  Neighbours n;
  std::memset(&n, 0, sizeof(n));
  n.x = static_cast<int16_t>(cu.GetPosX(luma));
  n.y = static_cast<int16_t>(cu.GetPosY(luma));
  n.w = static_cast<uint8_t>(cu.GetWidth(luma));
  n.h = static_cast<uint8_t>(cu.GetHeight(luma));
  const CodingUnit *above = cu.GetCodingUnitAbove(), *left = cu.GetCodingUnitLeft();
  n.has_above = above != nullptr;
  n.has_left = left != nullptr;
  if (above) {
    n.above_x = static_cast<int16_t>(above->GetPosX(luma));
    n.above_y = static_cast<int16_t>(above->GetPosY(luma));
  }
  if (left) {
    n.left_x = static_cast<int16_t>(left->GetPosX(luma));
    n.left_y = static_cast<int16_t>(left->GetPosY(luma));
  }
  std::vector<uint16_t> smp;
  for (int c = 0; c < 3; c++) {
    const YuvComponent comp = YuvComponent(c);
    const int x = cu.GetPosX(comp), y = cu.GetPosY(comp);
    const int w = cu.GetWidth(comp), h = cu.GetHeight(comp);
    if (above)
      for (int i = 0; i < w; i++) smp.push_back(*rec_pic.GetSamplePtr(comp, x + i, y - 1));
    if (left)
      for (int i = 0; i < h; i++) smp.push_back(*rec_pic.GetSamplePtr(comp, x - 1, y + i));
  }
  n.sample_count = static_cast<uint32_t>(smp.size());
  std::string key(reinterpret_cast<const char *>(&n), sizeof(n));
  key.append(reinterpret_cast<const char *>(smp.data()), smp.size() * sizeof(uint16_t));
  auto it = g_nb_index.find(key);
  if (it != g_nb_index.end()) return it->second;
  n.sample_off = static_cast<uint32_t>(g_nb_samples.size());
  g_nb_samples.insert(g_nb_samples.end(), smp.begin(), smp.end());
  const int idx = static_cast<int>(g_nb.size());
  g_nb.push_back(n);
  g_nb_index.emplace(key, idx);
  return idx;
}

static int InterContextIndex(const CodingUnit &cu, const SyntaxWriter &writer) {
  Contexts &ctx = const_cast<Contexts &>(writer.GetContexts());
  xvcgpu_inter_contexts c;
  std::memset(&c, 0, sizeof(c));
  c.merge_flag = ctx.inter_merge_flag[0].state_;
  c.inter_dir_bi = ctx.GetInterDirBiCtx(cu).state_;
  c.inter_dir_l = ctx.inter_dir[4].state_;
  c.affine_flag = ctx.GetAffineCtx(cu).state_;
  c.ref_idx[0] = ctx.inter_ref_idx[0].state_;
  c.ref_idx[1] = ctx.inter_ref_idx[1].state_;
  c.mvd[0] = ctx.inter_mvd[0].state_;
  c.mvd[1] = ctx.inter_mvd[1].state_;
  c.mvp_idx = ctx.inter_mvp_idx[0].state_;
  c.fullpel_mv = ctx.GetInterFullpelMvCtx(cu).state_;
  c.lic_flag = ctx.lic_flag[0].state_;
  const PictureData *pd = cu.GetPicData();
  c.flags = static_cast<uint8_t>(
      (pd->GetPredictionType() == PicturePredictionType::kBi ? XVC_ICTX_PIC_BI : 0) |
      (cu.CanUseAffine() ? XVC_ICTX_CAN_AFFINE : 0) |
      (pd->GetUseLocalIlluminationCompensation() ? XVC_ICTX_PIC_LIC : 0));
  c.num_refs[0] = static_cast<uint8_t>(pd->GetRefPicLists()->GetNumRefPics(RefPicList::kL0));
  c.num_refs[1] = static_cast<uint8_t>(pd->GetRefPicLists()->GetNumRefPics(RefPicList::kL1));
  c.frac_bits = static_cast<uint16_t>(writer.GetFractionalBits());
  std::string key(reinterpret_cast<const char *>(&c), sizeof(c));
  auto it = g_ictx_index.find(key);
  if (it != g_ictx_index.end()) return it->second;
  const int idx = static_cast<int>(g_ictx.size());
  g_ictx.push_back(c);
  g_ictx_index.emplace(key, idx);
  return idx;
}

static void PutMvd(int32_t out[2][2], const CodingUnit &cu, RefPicList l) {
  if (cu.GetUseAffine()) {
    for (int k = 0; k < 2; k++) {
      out[k][0] = cu.GetMvdAffine(k, l).x;
      out[k][1] = cu.GetMvdAffine(k, l).y;
    }
  } else {
    out[0][0] = cu.GetMvDelta(l).x;
    out[0][1] = cu.GetMvDelta(l).y;
  }
}

template <typename MV, size_t N>
void AfterMotionEst(InterSearch *is, CodingUnit *cu, const Qp &qp, RefPicList ref_list,
                    int ref_idx, bool bipred, const std::array<MV, N> &mvp_list, const MV *boot,
                    const MV &mv, Distortion dist, const SyntaxWriter &writer, int mvp_idx) {
  if (!Wanted(*cu)) return;
  const bool affine = std::is_same<MV, MotionVector3>::value;
  {
    /* the candidate's price: the two statements that follow the hooked one
     * (:557-558, SetMv and SetMvd) applied here first - the reference repeats them
     * with the same arguments - then the reference's own GetInterPredBits, which
     * works on a throw-away copy of the entropy coder */
    cu->SetMv(mv, ref_list);
    is->SetMvd(cu, ref_list, mvp_list[mvp_idx], mv);
    const YuvComponent luma = YuvComponent::kY;
    Cand c;
    std::memset(&c, 0, sizeof(c));
    c.poc = static_cast<int32_t>(cu->GetPicData()->GetPoc());
    c.x = static_cast<int16_t>(cu->GetPosX(luma));
    c.y = static_cast<int16_t>(cu->GetPosY(luma));
    c.w = static_cast<uint8_t>(cu->GetWidth(luma));
    c.h = static_cast<uint8_t>(cu->GetHeight(luma));
    c.kind = static_cast<uint8_t>(affine ? (bipred ? kAffineBi : kAffineUni) : (bipred ? kBi : 0));
    c.flags = static_cast<uint8_t>((cu->GetFullpelMv() ? kFlagFullpel : 0) |
                                   (cu->GetUseLic() ? kFlagLic : 0));
    c.list = static_cast<uint8_t>(ref_list);
    c.ref_idx = static_cast<int8_t>(ref_idx);
    c.reused = !bipred && ref_list == RefPicList::kL1 && is->same_poc_in_l0_mapping_[ref_idx] >= 0;
    c.mvp_idx = static_cast<uint8_t>(mvp_idx);
    c.inter_dir = static_cast<uint8_t>(cu->GetInterDir());
    c.other_ref_idx = -1;
    if (bipred) {
      const RefPicList other = ReferencePictureLists::Inverse(ref_list);
      c.other_ref_idx = static_cast<int8_t>(cu->GetRefIdx(other));
      c.other_mvp_idx = static_cast<uint8_t>(cu->GetMvpIdx(other));
      c.force_mvd_zero_other = cu->GetForceMvdZero(other);
      PutMvd(c.other_mvd, *cu, other);
    }
    PutMv(c.mv, mv);
    for (size_t k = 0; k < N && k < 2; k++) PutMv(c.mvp[k], mvp_list[k]);
    if (bipred) {
      c.start_mvp_idx = static_cast<uint8_t>(is->unipred_best_mvp_idx_[static_cast<int>(ref_list)][ref_idx]);
    } else {
      /* EvalStartMvp's choice (:493-496; also made for list-1 pictures that then re-use
       * list 0's result) asked again: a pure function of CU, predictors and pictures
       * (scratch prediction buffer) */
      SampleBufferStorage scratch(constants::kMaxBlockSize, constants::kMaxBlockSize);
      Distortion cost = 0;
      c.start_mvp_idx = static_cast<uint8_t>(is->EvalStartMvp<std::is_same<MV, MotionVector3>::value>(
          *cu, qp, mvp_list, *cu->GetRefPicLists()->GetRefPic(ref_list, ref_idx), &scratch, &cost));
    }
    c.dist = static_cast<uint32_t>(dist);
    c.bits = static_cast<uint32_t>(is->GetInterPredBits(*cu, writer));
    c.lambda16 = static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt()));
    c.ictx_index = InterContextIndex(*cu, writer);
    g_cands.push_back(c);
    xr_seq::Stamp(5);
  }
  if (!bipred && !affine) return;   /* the TZ searches are tools/gen_me_golden.py's */
  const int li = static_cast<int>(ref_list);
  if (!bipred && ref_list == RefPicList::kL1 && is->same_poc_in_l0_mapping_[ref_idx] >= 0)
    return;                         /* list 0's result reused (:536-542): no search */
  assert(is->encoder_settings_.bipred_refinement_iterations == 1);
  const YuvComponent luma = YuvComponent::kY;
  MeStep s;
  std::memset(&s, 0, sizeof(s));
  s.poc = static_cast<int32_t>(cu->GetPicData()->GetPoc());
  s.x = static_cast<int16_t>(cu->GetPosX(luma));
  s.y = static_cast<int16_t>(cu->GetPosY(luma));
  s.w = static_cast<uint8_t>(cu->GetWidth(luma));
  s.h = static_cast<uint8_t>(cu->GetHeight(luma));
  s.kind = static_cast<uint8_t>(affine ? (bipred ? kAffineBi : kAffineUni) : kBi);
  s.flags = static_cast<uint8_t>((cu->GetFullpelMv() ? kFlagFullpel : 0) |
                                 (cu->GetUseLic() ? kFlagLic : 0) | (boot ? kFlagHasBoot : 0));
  s.list = static_cast<uint8_t>(li);
  s.ref_idx = static_cast<int8_t>(ref_idx);
  const ReferencePictureLists *rpl = cu->GetRefPicLists();
  s.ref_poc = static_cast<int32_t>(rpl->GetRefPoc(ref_list, ref_idx));
  s.other_ref_idx = -1;
  s.other_ref_poc = -1;
  if (bipred) {
    const RefPicList other = ReferencePictureLists::Inverse(ref_list);
    s.other_ref_idx = static_cast<int8_t>(cu->GetRefIdx(other));
    s.other_ref_poc = static_cast<int32_t>(rpl->GetRefPoc(other, cu->GetRefIdx(other)));
    OtherMv(*cu, other, static_cast<const MV *>(nullptr), s.other_mv);
    /* the predictor the uni-directional search of this (list, picture) ended on
     * (:485-489; untouched by this call with one refinement iteration) */
    s.start_mvp_idx = static_cast<uint8_t>(is->unipred_best_mvp_idx_[li][ref_idx]);
  } else {
    /* the start predictor is EvalStartMvp's choice (:493-496); mvp_idx has been
     * overwritten by EvalFinalMvpIdx since: ask the reference again (a pure
     * function of CU, predictors and pictures; scratch prediction buffer) */
    SampleBufferStorage scratch(constants::kMaxBlockSize, constants::kMaxBlockSize);
    Distortion cost = 0;
    s.start_mvp_idx = static_cast<uint8_t>(is->EvalStartMvp<std::is_same<MV, MotionVector3>::value>(
        *cu, qp, mvp_list, *rpl->GetRefPic(ref_list, ref_idx), &scratch, &cost));
  }
  s.final_mvp_idx = static_cast<uint8_t>(cu->GetMvpIdx(ref_list));
  s.lambda16 = static_cast<uint32_t>(std::floor(65536.0 * qp.GetLambdaSqrt()));
  for (size_t k = 0; k < N && k < 2; k++) PutMv(s.mvp[k], mvp_list[k]);
  if (boot) PutMv(s.boot, *boot);
  PutMv(s.mv, mv);
  s.dist = static_cast<uint32_t>(dist);
  s.nb_index = bipred ? NeighbourIndex(*cu, is->rec_pic_) : -1;
  g_steps.push_back(s);
  xr_seq::Stamp(1);
}

void AfterLoadState(InterSearch *, CodingUnit *cu, const char *what) {
  if (!Wanted(*cu)) return;
  const int which = !std::strcmp(what, "state_bi") ? 0 : !std::strcmp(what, "state_l0") ? 1 :
                    !std::strcmp(what, "state_l1_unique_poc") ? 2 : -1;
  if (which < 0) return;
  const YuvComponent luma = YuvComponent::kY;
  Final f;
  std::memset(&f, 0, sizeof(f));
  f.poc = static_cast<int32_t>(cu->GetPicData()->GetPoc());
  f.x = static_cast<int16_t>(cu->GetPosX(luma));
  f.y = static_cast<int16_t>(cu->GetPosY(luma));
  f.w = static_cast<uint8_t>(cu->GetWidth(luma));
  f.h = static_cast<uint8_t>(cu->GetHeight(luma));
  f.which = static_cast<uint8_t>(which);
  f.flags = static_cast<uint8_t>((cu->GetFullpelMv() ? kFlagFullpel : 0) |
                                 (cu->GetUseLic() ? kFlagLic : 0) |
                                 (cu->GetUseAffine() ? kFlagAffine : 0));
  f.inter_dir = static_cast<uint8_t>(cu->GetInterDir());
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    const bool used = cu->GetInterDir() == InterDir::kBi || static_cast<int>(cu->GetInterDir()) == l;
    f.ref_idx[l] = static_cast<int8_t>(used ? cu->GetRefIdx(rl) : -1);
    if (!used) continue;
    f.mvp_idx[l] = static_cast<uint8_t>(cu->GetMvpIdx(rl));
    if (cu->GetUseAffine())
      PutMv(f.mv[l], cu->GetMvAffine(rl));
    else
      PutMv(f.mv[l], cu->GetMv(rl, MvCorner::kDefault));
    PutMvd(f.mvd[l], *cu, rl);
  }
  g_finals.push_back(f);
  xr_seq::Stamp(6);
}
template void AfterMotionEst<MotionVector, 2>(InterSearch *, CodingUnit *, const Qp &, RefPicList,
                                              int, bool, const std::array<MotionVector, 2> &,
                                              const MotionVector *, const MotionVector &,
                                              Distortion, const SyntaxWriter &, int);
template void AfterMotionEst<MotionVector3, 2>(InterSearch *, CodingUnit *, const Qp &,
                                               RefPicList, int, bool,
                                               const std::array<MotionVector3, 2> &,
                                               const MotionVector3 *, const MotionVector3 &,
                                               Distortion, const SyntaxWriter &, int);

void AfterMergeSort(InterSearch *is, CodingUnit *cu, const Qp &qp,
                    const InterMergeCandidateList &merge_list,
                    const std::array<std::pair<int, double>, 5> &cand_cost) {
  if (!Wanted(*cu)) return;
  const YuvComponent luma = YuvComponent::kY;
  MergeCall m;
  std::memset(&m, 0, sizeof(m));
  m.poc = static_cast<int32_t>(cu->GetPicData()->GetPoc());
  m.x = static_cast<int16_t>(cu->GetPosX(luma));
  m.y = static_cast<int16_t>(cu->GetPosY(luma));
  m.w = static_cast<uint8_t>(cu->GetWidth(luma));
  m.h = static_cast<uint8_t>(cu->GetHeight(luma));
  m.lambda_sqrt = qp.GetLambdaSqrt();
  const ReferencePictureLists *rpl = cu->GetRefPicLists();
  for (int k = 0; k < 5; k++) {
    const MergeCandidate &c = merge_list[k];
    m.inter_dir[k] = static_cast<uint8_t>(c.inter_dir);
    m.use_lic[k] = c.use_lic;
    for (int l = 0; l < 2; l++) {
      const bool used = c.inter_dir == InterDir::kBi || static_cast<int>(c.inter_dir) == l;
      m.ref_idx[k][l] = static_cast<int8_t>(used ? c.ref_idx[l] : -1);
      m.ref_poc[k][l] =
          used ? static_cast<int32_t>(rpl->GetRefPoc(static_cast<RefPicList>(l), c.ref_idx[l])) : -1;
      m.mv[k][l][0] = c.mv[l].x;
      m.mv[k][l][1] = c.mv[l].y;
    }
    m.order[k] = cand_cost[k].first;
    m.cost[k] = cand_cost[k].second;
  }
  /* the function's tail (:186-195) on the sorted costs */
  m.num = InterSearch::kFastMergeNumCand;
  for (int k = InterSearch::kFastMergeNumCand; k >= 0; k--)
    if (cand_cost[k].second > cand_cost[0].second * InterSearch::kFastMergeCostFactor) m.num = k;
  bool any_lic = false;
  for (int k = 0; k < 5; k++) any_lic |= m.use_lic[k] != 0;
  m.nb_index = any_lic ? NeighbourIndex(*cu, is->rec_pic_, true) : -1;
  g_merges.push_back(m);
  xr_seq::Stamp(2);
}

static int ContextIndexIn(const SyntaxWriter &writer, std::vector<xvcgpu_rdoq_contexts> *tab,
                          std::unordered_map<std::string, int> *index) {
  xvcgpu_rdoq_contexts c;
  StoreContexts(writer.GetContexts(), &c);
  std::string key(reinterpret_cast<const char *>(&c), sizeof(c));
  auto it = index->find(key);
  if (it != index->end()) return it->second;
  const int idx = static_cast<int>(tab->size());
  tab->push_back(c);
  index->emplace(key, idx);
  return idx;
}
static int ContextIndex(const SyntaxWriter &writer) {
  return ContextIndexIn(writer, &g_ctx, &g_ctx_index);
}

static int QpIndexIn(const Qp &qp, int bd, std::vector<QpParams> *tab,
                     std::unordered_map<std::string, int> *index);
static int QpIndex(const Qp &qp, int bd) { return QpIndexIn(qp, bd, &g_qps, &g_qp_index); }
static int QpIndexIn(const Qp &qp, int bd, std::vector<QpParams> *tab,
                     std::unordered_map<std::string, int> *index) {
  QpParams q;
  std::memset(&q, 0, sizeof(q));
  for (int c = 0; c < 3; c++) {
    const YuvComponent yc = YuvComponent(c);
    q.qp_raw[c] = static_cast<int8_t>(qp.GetQpRaw(yc));
    const double lam = qp.GetLambdaScaled(yc);
    const double inv_scale = qp.GetInvScale(yc);
    q.lambda[c] = static_cast<int64_t>(lam * (1 << 16) + 0.5);
    q.rd_factor[c] = static_cast<int64_t>(inv_scale * inv_scale / lam / 16 /
                                          (1ull << (2 * (bd - 8))) + 0.5);
    q.dist_weight[c] = qp.GetDistortionWeight(yc);
  }
  std::string key(reinterpret_cast<const char *>(&q), sizeof(q));
  auto it = index->find(key);
  if (it != index->end()) return it->second;
  const int idx = static_cast<int>(tab->size());
  tab->push_back(q);
  index->emplace(key, idx);
  return idx;
}

static int EvalIndex(const CodingUnit &cu, const Qp &qp, const SyntaxWriter &writer, int bd,
                     const YuvPicture &rec_pic) {
  const YuvComponent luma = YuvComponent::kY;
  Eval e;
  std::memset(&e, 0, sizeof(e));
  e.poc = static_cast<int32_t>(cu.GetPicData()->GetPoc());
  e.x = static_cast<int16_t>(cu.GetPosX(luma));
  e.y = static_cast<int16_t>(cu.GetPosY(luma));
  e.w = static_cast<uint8_t>(cu.GetWidth(luma));
  e.h = static_cast<uint8_t>(cu.GetHeight(luma));
  e.inter_dir = static_cast<uint8_t>(cu.GetInterDir());
  e.flags = static_cast<uint8_t>((cu.GetFullpelMv() ? kFlagFullpel : 0) |
                                 (cu.GetUseLic() ? kFlagLic : 0) |
                                 (cu.GetUseAffine() ? kFlagAffine : 0) |
                                 (cu.GetMergeFlag() ? kFlagMerge : 0));
  const ReferencePictureLists *rpl = cu.GetRefPicLists();
  for (int l = 0; l < 2; l++) {
    const RefPicList rl = static_cast<RefPicList>(l);
    const bool used = cu.GetInterDir() == InterDir::kBi || static_cast<int>(cu.GetInterDir()) == l;
    e.ref_idx[l] = static_cast<int8_t>(used ? cu.GetRefIdx(rl) : -1);
    e.ref_poc[l] = used ? static_cast<int32_t>(rpl->GetRefPoc(rl, cu.GetRefIdx(rl))) : -1;
    if (!used) continue;
    if (cu.GetUseAffine())
      PutMv(e.mv[l], cu.GetMvAffine(rl));
    else
      PutMv(e.mv[l], cu.GetMv(rl, MvCorner::kDefault));
  }
  for (int c = 0; c < 3; c++) {
    e.qp[c] = static_cast<int8_t>(qp.GetQpRaw(YuvComponent(c)));
    e.dist_zero[c] = ~0ull;
  }
  e.ctx_index = ContextIndex(writer);
  e.qp_index = QpIndex(qp, bd);
  e.nb_index = NeighbourIndex(cu, rec_pic);
  if (!g_evals.empty()) {
    /* the calls of one CompressAndEvalCbf share the CU state: same record */
    Eval last = g_evals.back();
    for (int c = 0; c < 3; c++) last.dist_zero[c] = ~0ull;
    if (std::memcmp(&last, &e, sizeof(e)) == 0) return static_cast<int>(g_evals.size()) - 1;
  }
  g_evals.push_back(e);
  xr_seq::Stamp(3);
  return static_cast<int>(g_evals.size()) - 1;
}

static void RecordIntraTx(TransformEncoder *te, CodingUnit *cu, YuvComponent comp, const Qp &qp,
                          const SyntaxWriter &writer, int non_zero, const YuvPicture &rec_pic,
                          const SampleBuffer &pred_buffer) {
  IntraMode mode = cu->GetIntraMode(comp);
  const bool lm = mode == IntraMode::kLmChroma;
  if (lm && g_itx_walk) mode = static_cast<IntraMode>(67);   /* XVC_INTRA_MODE_LM_CHROMA */
  if (static_cast<int>(mode) < 0 || static_cast<int>(mode) > (g_itx_walk ? 67 : 66)) return;
  if (g_itx_walk && g_itx_poc >= 0 && static_cast<int>(cu->GetPicData()->GetPoc()) != g_itx_poc)
    return;
  if ((g_itx_seen++ % g_itx_stride) != 0 || static_cast<int>(g_itx.size()) >= g_itx_cap) return;
  const int bd = te->max_pel_ == 1023 ? 10 : (te->max_pel_ == 255 ? 8 : 12);
  IntraTx t;
  std::memset(&t, 0, sizeof(t));
  t.poc = static_cast<int32_t>(cu->GetPicData()->GetPoc());
  t.x = static_cast<int16_t>(cu->GetPosX(comp));
  t.y = static_cast<int16_t>(cu->GetPosY(comp));
  t.w = static_cast<uint8_t>(cu->GetWidth(comp));
  t.h = static_cast<uint8_t>(cu->GetHeight(comp));
  t.comp = static_cast<uint8_t>(comp);
  t.mode = static_cast<uint8_t>(mode);
  /* IntraPrediction::DetermineNeighbors (intra_prediction.cc:688-705) */
  const bool has_left = t.x > 0, has_above = t.y > 0;
  t.neighbors = static_cast<uint8_t>((has_left && has_above ? XVC_INTRA_HAS_ABOVE_LEFT : 0) |
                                     (has_above ? XVC_INTRA_HAS_ABOVE : 0) |
                                     (has_left ? XVC_INTRA_HAS_LEFT : 0));
  t.above_right = static_cast<uint8_t>(has_above ? cu->GetCuSizeAboveRight(comp) : 0);
  t.below_left = static_cast<uint8_t>(has_left ? cu->GetCuSizeBelowLeft(comp) : 0);
  t.tx_skip = cu->GetTransformSkip(comp);
  t.tx_ver = static_cast<uint8_t>(cu->GetTransformType(comp, 0));
  t.tx_hor = static_cast<uint8_t>(cu->GetTransformType(comp, 1));
  t.scan = static_cast<uint8_t>(TransformHelper::DetermineScanOrder(*cu, comp));
  t.dst4x4 = util::IsLuma(comp) && t.tx_ver == 0 && t.tx_hor == 0;   /* transform.cc:88-90 */
  t.intra_pic = cu->GetPicType() == PicturePredictionType::kIntra;
  t.qp = static_cast<int8_t>(qp.GetQpRaw(comp));
  t.qp_luma = static_cast<int8_t>(qp.GetQpRaw(YuvComponent::kY));
  t.ctx_index = g_itx_walk ? ContextIndexIn(writer, &g_itx_ctx, &g_itx_ctx_index)
                           : ContextIndex(writer);
  t.qp_index = g_itx_walk ? QpIndexIn(qp, bd, &g_itx_qps, &g_itx_qp_index) : QpIndex(qp, bd);
  t.nnz = non_zero;
  CoeffBuffer coeff = cu->GetCoeff(comp);
  uint32_t crc = 0, pcrc = 0;
  for (int y = 0; y < t.h; y++) {
    crc = Crc32(crc, coeff.GetDataPtr() + y * coeff.GetStride(), sizeof(Coeff) * t.w);
    pcrc = Crc32(pcrc, pred_buffer.GetDataPtr() + y * pred_buffer.GetStride(),
                 sizeof(Sample) * t.w);
  }
  t.levels_crc = crc;
  t.pred_crc = pcrc;
  t.sample_off = static_cast<int32_t>(g_itx_samples.size());
  const Sample *p = rec_pic.GetSamplePtr(comp, t.x, t.y);
  const ptrdiff_t st = rec_pic.GetStride(comp);
  if (has_left && has_above) g_itx_samples.push_back(p[-st - 1]);
  if (has_above)
    for (int i = 0; i < t.w + t.above_right; i++) g_itx_samples.push_back(p[-st + i]);
  if (has_left)
    for (int i = 0; i < t.h + t.below_left; i++) g_itx_samples.push_back(p[i * st - 1]);
  if (lm) {   /* the luma rectangle of the linear model (the CU's own luma reconstruction) */
    const YuvComponent luma = YuvComponent::kY;
    const int lx = cu->GetPosX(luma), ly = cu->GetPosY(luma);
    const int lw = cu->GetWidth(luma), lh = cu->GetHeight(luma);
    const Sample *q = rec_pic.GetSamplePtr(luma, lx, ly);
    const ptrdiff_t ls = rec_pic.GetStride(luma);
    for (int y = (ly > 0 ? -2 : 0); y < lh; y++)
      for (int x = (lx > 0 ? -3 : 0); x < lw; x++) g_itx_samples.push_back(q[y * ls + x]);
  }
  g_itx.push_back(t);
  if (g_itx_walk) xr_seq::Stamp(8);
  g_pending_cu = cu;
  g_pending_itx = static_cast<int>(g_itx.size()) - 1;
}

void AfterQuantRdo(TransformEncoder *te, CodingUnit *cu, YuvComponent comp, const Qp &qp,
                   const SyntaxWriter &writer, int non_zero, const YuvPicture &rec_pic,
                   const SampleBuffer &pred_buffer) {
  g_pending_cu = nullptr;
  g_pending_itx = -1;
  if (!Wanted(*cu)) return;
  if (!cu->IsInter()) {
    g_skipped_intra++;
    if (g_itx_stride > 0) RecordIntraTx(te, cu, comp, qp, writer, non_zero, rec_pic, pred_buffer);
    return;
  }
  const int bd = te->max_pel_ == 1023 ? 10 : (te->max_pel_ == 255 ? 8 : 12);
  TxCall t;
  std::memset(&t, 0, sizeof(t));
  t.eval = EvalIndex(*cu, qp, writer, bd, rec_pic);
  t.comp = static_cast<uint8_t>(comp);
  t.tx_skip = cu->GetTransformSkip(comp);
  t.tx_ver = static_cast<uint8_t>(cu->GetTransformType(comp, 0));
  t.tx_hor = static_cast<uint8_t>(cu->GetTransformType(comp, 1));
  t.scan = static_cast<uint8_t>(TransformHelper::DetermineScanOrder(*cu, comp));
  t.tx_select_idx = static_cast<int8_t>(cu->GetTransformSelectIdx());
  t.nnz = non_zero;
  const int w = cu->GetWidth(comp), h = cu->GetHeight(comp);
  CoeffBuffer coeff = cu->GetCoeff(comp);
  uint32_t crc = 0;
  for (int y = 0; y < h; y++)
    crc = Crc32(crc, coeff.GetDataPtr() + y * coeff.GetStride(), sizeof(Coeff) * w);
  t.levels_crc = crc;
  g_calls.push_back(t);
  xr_seq::Stamp(4);
  g_pending_cu = cu;
  g_pending_call = static_cast<int>(g_calls.size()) - 1;
}

Distortion AfterCompare(TransformEncoder *te, CodingUnit *cu, YuvComponent comp,
                        const YuvPicture &orig_pic, const SampleBuffer &buffer, const char *func) {
  if (!Wanted(*cu)) return 0;
  const bool reconstruct = func[0] == 'T';   /* TransformAndReconstruct / CompressAndEvalTransform */
  if (!cu->IsInter()) {
    /* a sampled intra call: completed by its reconstruction and distortion */
    if (!reconstruct || g_pending_cu != cu || g_pending_itx < 0) return 0;
    IntraTx &t = g_itx[g_pending_itx];
    if (t.comp != static_cast<uint8_t>(comp)) return 0;
    uint32_t crc = 0;
    for (int y = 0; y < t.h; y++)
      crc = Crc32(crc, buffer.GetDataPtr() + y * buffer.GetStride(), sizeof(Sample) * t.w);
    t.rec_crc = crc;
    t.dist = te->cu_metric_.CompareSample(*cu, comp, orig_pic, buffer);
    t.completed = 1;
    g_pending_cu = nullptr;
    g_pending_itx = -1;
    return 0;
  }
  const Distortion d = te->cu_metric_.CompareSample(*cu, comp, orig_pic, buffer);
  if (!reconstruct) {
    /* cbf-zero distortion of the component: prediction against the original */
    if (!g_evals.empty()) {
      Eval &e = g_evals.back();
      const YuvComponent luma = YuvComponent::kY;
      if (e.x == cu->GetPosX(luma) && e.y == cu->GetPosY(luma) && e.w == cu->GetWidth(luma) &&
          e.h == cu->GetHeight(luma))
        e.dist_zero[static_cast<int>(comp)] = d;
    }
    return 0;
  }
  if (g_pending_cu != cu || g_pending_call < 0) return 0;
  TxCall &t = g_calls[g_pending_call];
  if (t.comp != static_cast<uint8_t>(comp)) return 0;
  const int w = cu->GetWidth(comp), h = cu->GetHeight(comp);
  uint32_t crc = 0;
  for (int y = 0; y < h; y++)
    crc = Crc32(crc, buffer.GetDataPtr() + y * buffer.GetStride(), sizeof(Sample) * w);
  t.rec_crc = crc;
  t.dist = d;
  t.completed = 1;
  g_pending_cu = nullptr;
  return 0;
}

}  // namespace xr_rd

/* ---- the intra search's SATD pre-selection, captured ------------------------ */
namespace xr_intra {

struct Call {                /* one DetermineSlowIntraModes */
  int32_t poc;
  int16_t x, y;              /* luma */
  uint8_t w, h;
  uint8_t neighbors;         /* XVC_INTRA_HAS_* */
  uint8_t above_right, below_left;   /* samples (DetermineNeighbors) */
  uint8_t pad[3];
  int32_t sample_off;        /* into the sample array: [above-left] [above: w + above_right]
                                [left: h + below_left], present parts only */
  int32_t first_eval, n_eval;
};
struct Eval {
  int32_t call;
  uint32_t dist;             /* SampleMetric(kSatd)::CompareSample */
  uint8_t mode;
  uint8_t pad[3];
};

bool g_capture = false;
int g_max_calls = 0, g_stride = 1;
int g_only_poc = -1;         /* round 5: only this picture's calls, stamped (xr_seq table 7) */
long g_seen = 0;             /* DetermineSlowIntraModes calls so far: every g_stride-th is kept */
std::vector<Call> g_calls;
std::vector<Eval> g_evals;
std::vector<uint16_t> g_samples;
bool g_open = false;         /* the evaluations that follow belong to g_calls.back() */

void OnCu(xvc::IntraSearch *is, xvc::CodingUnit *cu, const xvc::YuvPicture &rec_pic) {
  g_open = false;
  if (!g_capture) return;
  if (g_only_poc >= 0 && static_cast<int>(cu->GetPoc()) != g_only_poc) return;
  if ((g_seen++ % g_stride) != 0 || static_cast<int>(g_calls.size()) >= g_max_calls) return;
  const xvc::YuvComponent comp = xvc::YuvComponent::kY;
  const xvc::IntraPrediction::NeighborState nb = is->DetermineNeighbors(*cu, comp);
  Call c;
  std::memset(&c, 0, sizeof(c));
  c.poc = static_cast<int32_t>(cu->GetPoc());
  c.x = static_cast<int16_t>(cu->GetPosX(comp));
  c.y = static_cast<int16_t>(cu->GetPosY(comp));
  c.w = static_cast<uint8_t>(cu->GetWidth(comp));
  c.h = static_cast<uint8_t>(cu->GetHeight(comp));
  c.neighbors = static_cast<uint8_t>((nb.has_above_left ? XVC_INTRA_HAS_ABOVE_LEFT : 0) |
                                     (nb.has_above ? XVC_INTRA_HAS_ABOVE : 0) |
                                     (nb.has_left ? XVC_INTRA_HAS_LEFT : 0));
  c.above_right = static_cast<uint8_t>(nb.has_above_right);
  c.below_left = static_cast<uint8_t>(nb.has_below_left);
  c.sample_off = static_cast<int32_t>(g_samples.size());
  c.first_eval = static_cast<int32_t>(g_evals.size());
  const xvc::Sample *p = rec_pic.GetSamplePtr(comp, c.x, c.y);
  const ptrdiff_t st = rec_pic.GetStride(comp);
  if (nb.has_above_left) g_samples.push_back(p[-st - 1]);
  if (nb.has_above)
    for (int i = 0; i < c.w + c.above_right; i++) g_samples.push_back(p[-st + i]);
  if (nb.has_left)
    for (int i = 0; i < c.h + c.below_left; i++) g_samples.push_back(p[i * st - 1]);
  g_calls.push_back(c);
  if (g_only_poc >= 0) xr_seq::Stamp(7);
  g_open = true;
}

double OnEval(xvc::CodingUnit *, int mode, uint64_t dist) {
  if (g_open) {
    Eval e;
    std::memset(&e, 0, sizeof(e));
    e.call = static_cast<int32_t>(g_calls.size()) - 1;
    e.dist = static_cast<uint32_t>(dist);
    e.mode = static_cast<uint8_t>(mode);
    g_evals.push_back(e);
    g_calls.back().n_eval++;
  }
  return 0.0;
}

}  // namespace xr_intra

extern "C" {

void xr_intra_capture_begin(int max_calls, int stride) {
  xr_intra::g_stride = stride > 0 ? stride : 1;
  xr_intra::g_seen = 0;
  xr_intra::g_calls.clear();
  xr_intra::g_evals.clear();
  xr_intra::g_samples.clear();
  xr_intra::g_max_calls = max_calls;
  xr_intra::g_open = false;
  xr_intra::g_capture = true;
}
void xr_intra_capture_end(void) {
  xr_intra::g_capture = false;
  xr_intra::g_open = false;
  xr_intra::g_only_poc = -1;
}
/* after xr_intra_capture_begin: every call of picture `poc` only, in the global order */
void xr_intra_capture_poc(int poc) {
  xr_intra::g_only_poc = poc;
  xr_seq::g_of[7].clear();
}
/* which: 0 calls, 1 evaluations, 2 samples, 3 (count only) all calls seen */
long xr_intra_count(int which) {
  if (which == 3) return xr_intra::g_seen;
  return which == 0 ? static_cast<long>(xr_intra::g_calls.size())
         : which == 1 ? static_cast<long>(xr_intra::g_evals.size())
                      : static_cast<long>(xr_intra::g_samples.size());
}
int xr_intra_size(int which) {
  return which == 0 ? static_cast<int>(sizeof(xr_intra::Call))
         : which == 1 ? static_cast<int>(sizeof(xr_intra::Eval)) : 2;
}
const void *xr_intra_data(int which) {
  return which == 0 ? static_cast<const void *>(xr_intra::g_calls.data())
         : which == 1 ? static_cast<const void *>(xr_intra::g_evals.data())
                      : static_cast<const void *>(xr_intra::g_samples.data());
}

/* The intra-CU half of the capture: every `stride`-th TransformAndReconstruct of an
 * intra CU (at most `cap`), in the same run as xr_rd_capture_begin (call after it).
 * xr_rd_count / _size / _data: which 9 = IntraTx, 10 = their neighbour samples. */
void xr_rd_capture_intra(int stride, int cap) {
  xr_rd::g_itx_stride = stride;
  xr_rd::g_itx_cap = cap;
}
/* the walk's capture (after xr_rd_capture_begin(poc)): every TransformAndReconstruct of
 * the picture's intra CUs, LM chroma too, tables of their own (which 14 contexts, 15 QPs) */
void xr_rd_capture_intra_walk(int cap, int poc) {
  xr_rd::g_itx_stride = 1;
  xr_rd::g_itx_cap = cap;
  xr_rd::g_itx_walk = true;
  xr_rd::g_itx_poc = poc;
}

void xr_rd_capture_begin(int only_poc) {
  using namespace xr_rd;  // NOLINT
  g_itx.clear();
  g_itx_samples.clear();
  g_itx_walk = false;
  g_itx_poc = -1;
  g_itx_ctx.clear();
  g_itx_qps.clear();
  g_itx_ctx_index.clear();
  g_itx_qp_index.clear();
  g_itx_stride = 0;
  g_itx_seen = 0;
  g_pending_itx = -1;
  g_cands.clear();
  g_finals.clear();
  g_ictx.clear();
  g_ictx_index.clear();
  for (int t = 1; t < 9; t++) xr_seq::g_of[t].clear();
  g_steps.clear();
  g_merges.clear();
  g_evals.clear();
  g_qps.clear();
  g_calls.clear();
  g_ctx.clear();
  g_nb.clear();
  g_nb_samples.clear();
  g_ctx_index.clear();
  g_qp_index.clear();
  g_nb_index.clear();
  g_skipped_intra = 0;
  g_pending_cu = nullptr;
  g_only_poc = only_poc;
  g_capture = true;
}
void xr_rd_capture_end(void) { xr_rd::g_capture = false; }
/* which: 0 MeStep, 1 MergeCall, 2 Eval, 3 QpParams, 4 TxCall, 5 contexts, 6 Neighbours,
 * 7 their samples, 8 (count only) transform calls of intra CUs, not kept; round 4:
 * 11 Cand, 12 Final, 13 inter-prediction context snapshots, 20 + t: the global
 * sequence numbers of table t (xr_seq: 0 me calls ... 6 finals) */
long xr_rd_count(int which) {
  using namespace xr_rd;  // NOLINT
  switch (which) {
    case 0: return static_cast<long>(g_steps.size());
    case 1: return static_cast<long>(g_merges.size());
    case 2: return static_cast<long>(g_evals.size());
    case 3: return static_cast<long>(g_qps.size());
    case 4: return static_cast<long>(g_calls.size());
    case 5: return static_cast<long>(g_ctx.size());
    case 6: return static_cast<long>(g_nb.size());
    case 7: return static_cast<long>(g_nb_samples.size());
    case 8: return g_skipped_intra;
    case 9: return static_cast<long>(g_itx.size());
    case 10: return static_cast<long>(g_itx_samples.size());
    case 11: return static_cast<long>(g_cands.size());
    case 12: return static_cast<long>(g_finals.size());
    case 13: return static_cast<long>(g_ictx.size());
    case 14: return static_cast<long>(g_itx_ctx.size());
    case 15: return static_cast<long>(g_itx_qps.size());
  }
  if (which >= 20 && which < 29) return static_cast<long>(xr_seq::g_of[which - 20].size());
  return -1;
}
int xr_rd_size(int which) {
  using namespace xr_rd;  // NOLINT
  switch (which) {
    case 0: return sizeof(MeStep);
    case 1: return sizeof(MergeCall);
    case 2: return sizeof(Eval);
    case 3: return sizeof(QpParams);
    case 4: return sizeof(TxCall);
    case 5: return sizeof(xvcgpu_rdoq_contexts);
    case 6: return sizeof(Neighbours);
    case 7: return sizeof(uint16_t);
    case 9: return sizeof(IntraTx);
    case 10: return sizeof(uint16_t);
    case 11: return sizeof(Cand);
    case 12: return sizeof(Final);
    case 13: return sizeof(xvcgpu_inter_contexts);
    case 14: return sizeof(xvcgpu_rdoq_contexts);
    case 15: return sizeof(QpParams);
  }
  if (which >= 20 && which < 29) return sizeof(uint32_t);
  return -1;
}
const void *xr_rd_data(int which) {
  using namespace xr_rd;  // NOLINT
  switch (which) {
    case 0: return g_steps.data();
    case 1: return g_merges.data();
    case 2: return g_evals.data();
    case 3: return g_qps.data();
    case 4: return g_calls.data();
    case 5: return g_ctx.data();
    case 6: return g_nb.data();
    case 7: return g_nb_samples.data();
    case 9: return g_itx.data();
    case 10: return g_itx_samples.data();
    case 11: return g_cands.data();
    case 12: return g_finals.data();
    case 13: return g_ictx.data();
    case 14: return g_itx_ctx.data();
    case 15: return g_itx_qps.data();
  }
  if (which >= 20 && which < 29) return xr_seq::g_of[which - 20].data();
  return nullptr;
}

/* ContextModel's state transitions (context_model.cc:51-73): the product builds its
 * own tables from the CABAC state machine's rule; the tests compare. */
const uint8_t *xr_next_state_table(int lps) {
  return lps ? &ContextModel::kNextStateLps_[0] : &ContextModel::kNextStateMps_[0];
}

}  // extern "C"
