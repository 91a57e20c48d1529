// host_demo.cc -- compiles the C++ host layer (xvc_amd/host/xvc_gpu_ops.h)
// against libxvcgpu.so and exercises it: PadBorder + SAD through the
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
    return d[0] == expect ? 0 : 1;
  } catch (const xvc_gpu::Error &e) {
    std::printf("xvc_gpu error %d: %s\n", static_cast<int>(e.status), e.what());
    return e.status == XVCGPU_NO_DEVICE ? 3 : 2;
  }
}
