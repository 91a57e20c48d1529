// xvc_frame_pass.h -- C++ host driver of the hot-path frame pass (the C++ twin
// of xvc_amd/pipeline.py::FramePass): what PictureEncoder::Encode
// (picture_encoder.cc:75-160) drives per picture once mode decision is taken
// out - for every CU the motion search (inter_search.cc:606-662) and
// CompressAndEvalCbf (:261-365), then DeblockPicture, PadBorder and the PSNR
// walk - all through the C-ABI, every picture and decision resident in HBM.
#ifndef XVC_AMD_HOST_XVC_FRAME_PASS_H_
#define XVC_AMD_HOST_XVC_FRAME_PASS_H_

#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "xvc_gpu_ops.h"

namespace xvc_gpu {

// Qp::ScaleChromaQp for 4:2:0 with chroma table 1 (quantize.cc:34-38, :74-82):
// identity below 30, then the standard chroma mapping.
inline int ChromaQp(int qp) {
  static const int8_t tail[] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  qp = qp < 0 ? 0 : (qp > 57 ? 57 : qp);
  if (qp < 30) return qp;
  if (qp < 44) return tail[qp - 30];
  return qp - 6;
}
// floor(65536 * sqrt(lambda)), lambda = 0.57 * 2^((qp-12)/3)
// (picture_data.cc:96-97, inter_tz_search.cc:98-99)
inline uint32_t Lambda16(int qp) {
  return static_cast<uint32_t>(
      std::floor(65536.0 * std::sqrt(0.57 * std::pow(2.0, (qp - 12) / 3.0))));
}

class FramePass {
 public:
  FramePass(const Context &ctx, int width, int height, int bitdepth, int qp,
            int cu = 16, int search_range = 96)
      : ctx_(ctx), w_(width), h_(height), bd_(bitdepth), qp_(qp), qp_c_(ChromaQp(qp)) {
    map_stride_ = (width + 3) / 4;
    std::vector<int32_t> map(static_cast<size_t>(map_stride_) * ((height + 3) / 4), -1);
    std::vector<xvcgpu_me_block> me;
    for (int y = 0; y < height; y += cu)
      for (int x = 0; x < width; x += cu) {
        xvcgpu_me_block b = xvcgpu_me_block();
        b.x = static_cast<int16_t>(x);
        b.y = static_cast<int16_t>(y);
        b.w = static_cast<uint8_t>(width - x < cu ? width - x : cu);
        b.h = static_cast<uint8_t>(height - y < cu ? height - y : cu);
        b.depth_nonzero = 1;
        b.lambda16 = Lambda16(qp);
        b.search_range = search_range;
        for (int yy = y / 4; yy < (y + b.h) / 4; yy++)
          for (int xx = x / 4; xx < (x + b.w) / 4; xx++)
            map[static_cast<size_t>(yy) * map_stride_ + xx] = static_cast<int32_t>(me.size());
        me.push_back(b);
      }
    n_cus_ = static_cast<int>(me.size());
    d_me_.reset(new DeviceArray<xvcgpu_me_block>(ctx, me));
    d_map_.reset(new DeviceArray<int32_t>(ctx, map));
    d_res_.reset(new DeviceArray<xvcgpu_me_result>(ctx, me.size()));
    d_nnz_.reset(new DeviceArray<int32_t>(ctx, 3 * me.size()));
    d_cus_.reset(new DeviceArray<xvcgpu_cu_info>(ctx, me.size()));
    d_ssd_.reset(new DeviceArray<uint64_t>(ctx, 2));
    ctx_.Check(xvcgpu_memset(ctx_.get(), d_cus_->data(), 0, me.size() * sizeof(xvcgpu_cu_info)));
    max_cu_ = cu;
  }

  // Enqueues one picture (asynchronous): rec becomes the padded reconstruction.
  // The whole sequence - search, CompressAndEvalCbf, deblocking, PadBorder,
  // PSNR parts - goes through one C call (xvcgpu_frame_pass).
  void Run(const Picture &orig, const Picture &ref, Picture *rec, int ref_poc = 0) {
    xvcgpu_frame_pass_args a = xvcgpu_frame_pass_args();
    a.orig = orig.get();
    a.ref = ref.get();
    a.rec = rec->get();
    a.d_me = d_me_->data();
    a.d_results = d_res_->data();
    a.n_cus = n_cus_;
    a.max_block_size = max_cu_;
    a.qp_y = qp_;
    a.qp_c = qp_c_;
    a.ref_poc = ref_poc;
    a.d_nnz = d_nnz_->data();
    a.d_cus_own = d_cus_->data();
    a.d_cus = d_cus_->data();
    a.n_cus_total = n_cus_;
    a.d_cu_map = d_map_->data();
    a.map_stride = map_stride_;
    a.db_y_begin = 0;
    a.db_y_end = a.dbh_y_end = h_;
    a.ssd_y_begin = 0;
    a.ssd_y_end = 1 << 30;
    a.shift_bitdepth = bd_;
    a.d_ssd = d_ssd_->data();
    ctx_.Check(xvcgpu_frame_pass(ctx_.get(), &a,
                                 XVC_FP_ENCODE | XVC_FP_DEBLOCK_V | XVC_FP_DEBLOCK_H |
                                     XVC_FP_PAD | XVC_FP_SSD));
  }

  // SampleMetric::ComputePsnr parts of the last Run (synchronises).
  void Ssd(uint64_t *ssd, uint64_t *samples) const {
    std::vector<uint64_t> v = d_ssd_->ToHost();
    *ssd = v[0];
    *samples = v[1];
  }
  std::vector<xvcgpu_me_result> MotionVectors() const { return d_res_->ToHost(); }
  int num_cus() const { return n_cus_; }

 private:
  const Context &ctx_;
  int w_, h_, bd_, qp_, qp_c_, n_cus_, map_stride_, max_cu_;
  std::unique_ptr<DeviceArray<xvcgpu_me_block>> d_me_;
  std::unique_ptr<DeviceArray<int32_t>> d_map_;
  std::unique_ptr<DeviceArray<xvcgpu_me_result>> d_res_;
  std::unique_ptr<DeviceArray<int32_t>> d_nnz_;
  std::unique_ptr<DeviceArray<xvcgpu_cu_info>> d_cus_;
  std::unique_ptr<DeviceArray<uint64_t>> d_ssd_;
};

}  // namespace xvc_gpu
#endif  // XVC_AMD_HOST_XVC_FRAME_PASS_H_
