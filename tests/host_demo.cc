// host_demo.cc -- compiles the C++ host layer (xvc_amd/host/xvc_gpu_ops.h)
// against libxvcgpu.so and exercises it: PadBorder + SAD, the sample
// conversions, the CRC, the AQP statistic and the LIC test through the
// reference-named classes.  Exit 0 = OK, 3 = no gfx950 device (no fallback).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "xvc_gpu_ops.h"

int main() {
  try {
    xvc_gpu::Context ctx(0);
    const int w = 64, h = 32, bd = 10;
    xvc_gpu::Picture a(ctx, w, h, bd), b(ctx, w, h, bd);
    std::vector<uint16_t> y(w * h), u(w * h / 4, 512), v(w * h / 4, 512), y2(w * h);
    for (int i = 0; i < w * h; i++) {
      y[i] = static_cast<uint16_t>((i * 37) & 1023);
      y2[i] = static_cast<uint16_t>((i * 37 + 5) & 1023);
    }
    const uint16_t *pa[3] = {y.data(), u.data(), v.data()};
    const uint16_t *pb[3] = {y2.data(), u.data(), v.data()};
    const ptrdiff_t st[3] = {w, w / 2, w / 2};
    a.Upload(pa, st);
    b.Upload(pb, st);
    a.PadBorder();
    b.PadBorder();
    xvc_gpu::SampleMetric metric(ctx);
    std::vector<xvcgpu_metric_cand> cands(1);
    cands[0].x = 16; cands[0].y = 8; cands[0].w = 16; cands[0].h = 16;
    cands[0].metric = XVC_METRIC_SAD; cands[0].qp = 32; cands[0].mv_x = 0; cands[0].mv_y = 0;
    std::vector<uint64_t> d = metric.CompareBatch(a, b, 0, 1.0, cands);
    // expected SAD computed here on the host from the inputs
    long sad = 0;
    for (int yy = 8; yy < 24; yy++)
      for (int xx = 16; xx < 32; xx++)
        sad += std::labs(static_cast<long>(y[yy * w + xx]) - y2[yy * w + xx]);
    const unsigned long long expect = static_cast<unsigned long long>(sad) >> (bd - 8);
    std::printf("sad %llu expect %llu\n", static_cast<unsigned long long>(d[0]), expect);
    if (d[0] != expect) return 1;

    // 8-bit application picture -> internal 10 bit -> back (both down-shifts)
    std::vector<uint8_t> in8(w * h * 3 / 2), out8;
    for (size_t i = 0; i < in8.size(); i++) in8[i] = static_cast<uint8_t>((i * 29 + 7) & 255);
    xvc_gpu::Resampler resampler(ctx);
    xvc_gpu::Picture p(ctx, w, h, bd);
    resampler.ConvertFrom(w, h, 8, in8.data(), &p);
    for (int dither = 0; dither < 2; dither++) {
      resampler.ConvertTo(p, w, h, 8, dither != 0, &out8);
      if (out8 != in8) return 4;
    }
    // CRC of the internal picture against the bit-serial definition
    xvc_gpu::Checksum checksum(ctx, xvc_gpu::Checksum::Mode::kMinOverhead);
    checksum.HashPicture(p);
    unsigned crc = 0xffff;
    for (size_t i = 0; i < in8.size() + 1; i++) {
      const unsigned sample = i < in8.size() ? static_cast<unsigned>(in8[i]) << 2 : 0;
      for (int bit = 0; bit < 16; bit++) {   // low byte first, each MSB first
        const unsigned msb = (crc >> 15) & 1;
        const unsigned b = i < in8.size() ? (sample >> ((bit < 8 ? 7 : 23) - bit)) & 1 : 0;
        crc = (((crc << 1) + b) & 0xffff) ^ (msb * 0x1021);
      }
    }
    const std::vector<uint8_t> hash = checksum.GetHash();
    std::printf("crc %02x%02x expect %04x\n", hash[0], hash[1], crc);
    if (hash.size() != 2 || hash[0] != (crc >> 8) || hash[1] != (crc & 0xff)) return 5;
    // AQP offsets (one per 16x16 CTU here) and the LIC test
    xvc_gpu::AdaptiveQp aqp(ctx, 13);
    const std::vector<int> dqp = aqp.CalcDeltaQpFromVariance(p, w, h, bd, 16);
    if (dqp.size() != static_cast<size_t>((w / 16) * (h / 16))) return 6;
    for (size_t i = 0; i < dqp.size(); i++)
      if (dqp[i] < -3 || dqp[i] > 7) return 6;
    if (xvc_gpu::DetermineAllowLic(ctx, p, p, w, h)) return 7;
    if (!xvc_gpu::DetermineAllowLic(ctx, p, a, w, h)) return 7;
    // intra: the vertical mode (50) copies the row above the block; the SATD
    // table of a block whose original equals that prediction is 0 at mode 50
    std::vector<xvcgpu_intra_block> ib(1);
    ib[0].x = 16; ib[0].y = 8; ib[0].w = 8; ib[0].h = 8; ib[0].comp = 0; ib[0].mode = 50;
    ib[0].neighbors = XVC_INTRA_HAS_ABOVE | XVC_INTRA_HAS_LEFT | XVC_INTRA_HAS_ABOVE_LEFT;
    ib[0].above_right = 8; ib[0].below_left = 8; ib[0].reserved = 0;
    xvc_gpu::Picture pred(ctx, w, h, bd);
    xvc_gpu::IntraPrediction(ctx).PredictBatch(a, &pred, ib);
    std::vector<uint16_t> py(w * h), pu(w * h / 4), pv(w * h / 4);
    uint16_t *pp[3] = {py.data(), pu.data(), pv.data()};
    pred.Download(pp, st);
    for (int yy = 8; yy < 16; yy++)       // (column 16 carries the edge filter)
      for (int xx = 17; xx < 24; xx++)
        if (py[yy * w + xx] != y[7 * w + xx]) return 8;
    const std::vector<uint32_t> dist = xvc_gpu::IntraSearch(ctx).SatdAllModesBatch(pred, a, ib);
    if (dist.size() != XVC_INTRA_NUM_MODES || dist[50] != 0 || dist[18] == 0) return 9;
    // affine ME of a picture against itself from the zero predictor: the
    // prediction equals the original, the normal equations have a zero
    // right-hand side, the first gradient step is zero and the search stops
    std::vector<xvcgpu_affine_me_block> aj(1);
    aj[0] = xvcgpu_affine_me_block();
    aj[0].x = 16; aj[0].y = 8; aj[0].w = 16; aj[0].h = 16; aj[0].lambda16 = 100000;
    const std::vector<xvcgpu_affine_me_result> ar =
        xvc_gpu::InterSearch(ctx).MotionEstAffineBatch(a, a, a, aj);
    if (ar.size() != 1 || ar[0].dist != 0 || ar[0].iterations != 0) return 10;
    for (int i = 0; i < 3; i++)
      if (ar[0].mv[i][0] != 0 || ar[0].mv[i][1] != 0) return 10;
    return 0;
  } catch (const xvc_gpu::Error &e) {
    std::printf("xvc_gpu error %d: %s\n", static_cast<int>(e.status), e.what());
    return e.status == XVCGPU_NO_DEVICE ? 3 : 2;
  }
}
