// frame_pass_main.cc -- C++ host program: N chained hot-path frame passes over
// the deterministic synthetic clip, through xvc_gpu::FramePass / the C-ABI.
// Prints one line per picture (SSD as ComputePsnr sums it, FNV-1a of the three
// reconstructed planes) and the throughput; tests/test_abi.py compares the
// lines with the oracle.
//   gcc -O2 -c xvc_amd/host/xvc_synth.c -o synth.o
//   g++ -std=c++11 -O2 -Iinclude -Ixvc_amd/host xvc_amd/host/frame_pass_main.cc synth.o
//       -Lxvc_amd -lxvcgpu -o frame_pass && ./frame_pass 1920 1080 10 32 8
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "xvc_frame_pass.h"

extern "C" void xo_synth_frame(int width, int height, int bitdepth, uint64_t seed, int square,
                               int n, uint16_t *Y, ptrdiff_t ys, uint16_t *U, ptrdiff_t us,
                               uint16_t *V, ptrdiff_t vs);

static uint64_t Fnv1a(const std::vector<uint16_t> &v, uint64_t h) {
  for (size_t i = 0; i < v.size(); i++) {
    h = (h ^ (v[i] & 0xff)) * 1099511628211ull;
    h = (h ^ (v[i] >> 8)) * 1099511628211ull;
  }
  return h;
}

int main(int argc, char **argv) {
  const int w = argc > 1 ? std::atoi(argv[1]) : 352, h = argc > 2 ? std::atoi(argv[2]) : 288;
  const int bd = argc > 3 ? std::atoi(argv[3]) : 10, qp = argc > 4 ? std::atoi(argv[4]) : 32;
  const int frames = argc > 5 ? std::atoi(argv[5]) : 3;
  try {
    xvc_gpu::Context ctx(0);
    std::vector<uint16_t> y(static_cast<size_t>(w) * h), u(y.size() / 4), v(y.size() / 4);
    uint16_t *planes[3] = {y.data(), u.data(), v.data()};
    const uint16_t *cplanes[3] = {y.data(), u.data(), v.data()};
    const ptrdiff_t strides[3] = {w, w / 2, w / 2};
    xvc_gpu::Picture orig(ctx, w, h, bd), rec_a(ctx, w, h, bd), rec_b(ctx, w, h, bd);
    xvc_gpu::Picture *ref = &rec_a, *rec = &rec_b;
    xo_synth_frame(w, h, bd, 1234, 1, 0, y.data(), w, u.data(), w / 2, v.data(), w / 2);
    ref->Upload(cplanes, strides);
    ref->PadBorder();
    xvc_gpu::FramePass fp(ctx, w, h, bd, qp);
    double gpu_s = 0;
    for (int n = 1; n <= frames; n++) {
      xo_synth_frame(w, h, bd, 1234, 1, n, y.data(), w, u.data(), w / 2, v.data(), w / 2);
      orig.Upload(cplanes, strides);
      ctx.Sync();
      const auto t0 = std::chrono::steady_clock::now();
      fp.Run(orig, *ref, rec, n - 1);
      ctx.Sync();
      gpu_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      uint64_t ssd = 0, samples = 0;
      fp.Ssd(&ssd, &samples);
      rec->Download(planes, strides);
      uint64_t hash = 14695981039346656037ull;
      hash = Fnv1a(y, hash);
      hash = Fnv1a(u, hash);
      hash = Fnv1a(v, hash);
      std::printf("frame %d ssd %llu samples %llu fnv %016llx\n", n,
                  static_cast<unsigned long long>(ssd),
                  static_cast<unsigned long long>(samples),
                  static_cast<unsigned long long>(hash));
      std::swap(ref, rec);
    }
    std::printf("%d CUs per picture, %.1f frame passes/s (one launch sequence at a time)\n",
                fp.num_cus(), frames / gpu_s);
    return 0;
  } catch (const xvc_gpu::Error &e) {
    std::printf("xvc_gpu error %d: %s\n", static_cast<int>(e.status), e.what());
    return e.status == XVCGPU_NO_DEVICE ? 3 : 2;
  }
}
