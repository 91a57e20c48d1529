// k_metric.h -- M1..M7: SampleMetric::Compare (sample_metric.cc:171-223) as
// device functions over one wavefront, and the batched candidate kernel.
//
// All functions are called by a full 64-lane wave with wave-uniform arguments
// and return the (wave-uniform) metric value before the `weight` scaling.
// Samples are fetched with 2-byte loads (candidate blocks sit at arbitrary
// full-pel offsets, i.e. 2-byte alignment); consecutive lanes read
// consecutive samples of a row, so every wave instruction is one or two
// contiguous segments.  Integer sums are int32 as in the reference's C
// kernels; wave reductions use xor-shuffles (DPP on gfx950).
#ifndef XVCGPU_K_METRIC_H_
#define XVCGPU_K_METRIC_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

// Typed sample accessor so the same code serves Sample and Residual operands.
template <typename T>
__device__ __forceinline__ int ld_s(const T *p, ptrdiff_t i) {
  return (int)p[i];
}

// ---- SAD (ComputeSad_c, sample_metric.cc:670-684) --------------------------
// rows: number of rows visited, row_step: 1 (kSad) or 2 (kSadFast).
template <typename T1>
__device__ __forceinline__ int wave_sad(int w, int rows, int row_step,
                                        const T1 *a, int sa, const uint16_t *b,
                                        int sb) {
  const int lane = threadIdx.x & 63;
  const int n = w * rows;
  const int lw = 31 - __clz(w);  // w is a power of two
  int sum = 0;
  for (int i = lane; i < n; i += 64) {
    const int y = (i >> lw) * row_step, x = i & (w - 1);
    sum += d_abs(ld_s(a, (ptrdiff_t)y * sa + x) - (int)b[(ptrdiff_t)y * sb + x]);
  }
  return group_sum<64>(sum);
}

// ---- SSD (ComputeSsd_c, sample_metric.cc:300-314) --------------------------
template <typename T1>
__device__ __forceinline__ uint64_t wave_ssd(int w, int h, const T1 *a, int sa,
                                             const uint16_t *b, int sb) {
  const int lane = threadIdx.x & 63;
  const int n = w * h;
  const int lw = 31 - __clz(w);
  uint64_t sum = 0;
  for (int i = lane; i < n; i += 64) {
    const int y = i >> lw, x = i & (w - 1);
    const int d = ld_s(a, (ptrdiff_t)y * sa + x) - (int)b[(ptrdiff_t)y * sb + x];
    sum += (uint64_t)(int64_t)(d * d);
  }
  return group_sum<64>(sum);
}

// ---- mean difference (CalcMeanDiff, sample_metric.cc:769-783) --------------
template <typename T1>
__device__ __forceinline__ int wave_mean_diff(int skip, int w, int h,
                                              const T1 *a, int sa,
                                              const uint16_t *b, int sb) {
  const int lane = threadIdx.x & 63;
  const int step = 1 + skip;
  const int rows = (h + step - 1) / step;
  const int n = w * rows;
  const int lw = 31 - __clz(w);
  int sum = 0;
  for (int i = lane; i < n; i += 64) {
    const int y = (i >> lw) * step, x = i & (w - 1);
    sum += ld_s(a, (ptrdiff_t)y * sa + x) - (int)b[(ptrdiff_t)y * sb + x];
  }
  sum = group_sum<64>(sum);
  return (sum * step) / (w * h);  // C division: truncates toward zero
}

// ---- SAD AC-only (ComputeSadAcOnly, sample_metric.cc:686-703) --------------
template <typename T1>
__device__ __forceinline__ uint64_t wave_sad_ac(int skip, int bd, int w, int h,
                                                const T1 *a, int sa,
                                                const uint16_t *b, int sb) {
  const int avg = wave_mean_diff(skip, w, h, a, sa, b, sb);
  const int lane = threadIdx.x & 63;
  const int step = 1 + skip;
  const int rows = (h + step - 1) / step;
  const int n = w * rows;
  const int lw = 31 - __clz(w);
  int sum = 0;
  for (int i = lane; i < n; i += 64) {
    const int y = (i >> lw) * step, x = i & (w - 1);
    sum += d_abs(ld_s(a, (ptrdiff_t)y * sa + x) - (int)b[(ptrdiff_t)y * sb + x] -
                 avg);
  }
  sum = group_sum<64>(sum);
  return (uint64_t)(int64_t)((sum * step) >> (bd - 8));
}

// ---- SATD (ComputeSatd / ComputeSatdNxM, sample_metric.cc:316-668) ---------
// A TW x TH Hadamard tile is held by TH lanes (one row each, TW registers).
// The horizontal WHT runs in registers, the vertical WHT across the TH lanes
// with xor-shuffles; sum|.| is invariant to the butterfly order so this is
// bit-identical to the reference's fixed butterfly network.  64/TH tiles are
// processed per wave iteration.
template <int TW, int TH, typename T1>
__device__ __forceinline__ uint64_t wave_satd_tiles(int w, int h, int off,
                                                    const T1 *a, int sa,
                                                    const uint16_t *b, int sb) {
  const int lane = threadIdx.x & 63;
  constexpr int TPW = 64 / TH;  // tiles per wave iteration
  const int row = lane % TH, slot = lane / TH;
  const int tiles_x = w / TW, n_tiles = tiles_x * (h / TH);
  int total = 0;  // accumulated by the row-0 lane of each slot
  for (int t0 = 0; t0 < n_tiles; t0 += TPW) {
    const int t = t0 + slot;
    const bool active = t < n_tiles;
    int m[TW];
    if (active) {
      const int tx = (t % tiles_x) * TW, ty = (t / tiles_x) * TH + row;
      const T1 *pa = a + (ptrdiff_t)ty * sa + tx;
      const uint16_t *pb = b + (ptrdiff_t)ty * sb + tx;
#pragma unroll
      for (int x = 0; x < TW; x++) m[x] = ld_s(pa, x) - (int)pb[x] - off;
    } else {
#pragma unroll
      for (int x = 0; x < TW; x++) m[x] = 0;
    }
#pragma unroll
    for (int len = 1; len < TW; len <<= 1)
#pragma unroll
      for (int i = 0; i < TW; i += len << 1)
#pragma unroll
        for (int j = i; j < i + len; j++) {
          const int u = m[j], v = m[j + len];
          m[j] = u + v;
          m[j + len] = u - v;
        }
#pragma unroll
    for (int s = 1; s < TH; s <<= 1) {
      const bool upper = (row & s) != 0;
#pragma unroll
      for (int x = 0; x < TW; x++) {
        const int o = __shfl_xor(m[x], s, XVC_WAVE);
        m[x] = upper ? o - m[x] : m[x] + o;
      }
    }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < TW; x++) sum += d_abs(m[x]);
    sum = group_sum<TH>(sum);
    if (TW == 2 && TH == 2) {
      // ComputeSatd2x2: no normalisation
    } else if (TW == 4 && TH == 4) {
      sum = (sum + 1) >> 1;
    } else if (TW == TH) {
      sum = (sum + 2) >> 2;
    } else {
      // static_cast<int>(2.0 * sum / std::sqrt(W*H)), sample_metric.cc:638
      sum = (int)(2.0 * (double)sum / sqrt((double)(TW * TH)));
    }
    if (active && row == 0) total += sum;
  }
  return (uint64_t)(int64_t)group_sum<64>(total);
}

template <typename T1>
__device__ __forceinline__ uint64_t wave_satd(int bd, int w, int h, int off,
                                              const T1 *a, int sa,
                                              const uint16_t *b, int sb) {
  uint64_t sad;
  if (w == 2 || h == 2) {
    sad = wave_satd_tiles<2, 2>(w, h, off, a, sa, b, sb);
  } else if (w == 4 && h == 4) {
    sad = wave_satd_tiles<4, 4>(w, h, off, a, sa, b, sb);
  } else if (h == 4 && w > h) {
    sad = wave_satd_tiles<8, 4>(w, h, off, a, sa, b, sb);
  } else if (w == 4 && h > w) {
    sad = wave_satd_tiles<4, 8>(w, h, off, a, sa, b, sb);
  } else if (w > h) {
    sad = wave_satd_tiles<16, 8>(w, h, off, a, sa, b, sb);
  } else if (w < h) {
    sad = wave_satd_tiles<8, 16>(w, h, off, a, sa, b, sb);
  } else {
    sad = wave_satd_tiles<8, 8>(w, h, off, a, sa, b, sb);
  }
  return sad >> (bd - 8);
}

// ---- structural SSD (sample_metric.cc:705-767) ------------------------------
// One size x size sub-block per group of size*size lanes; the group leader
// does the ~15 double operations in the reference's exact order (compiled
// with -ffp-contract=off; device double +,-,*,/ are IEEE correctly rounded).
template <typename T1>
__device__ __forceinline__ uint64_t wave_structural_ssd(int bd, int qp_raw,
                                                        int strength, int w,
                                                        int h, const T1 *a,
                                                        int sa,
                                                        const uint16_t *b,
                                                        int sb) {
  const int lane = threadIdx.x & 63;
  const int size = (h < 8 || w < 8) ? 4 : 8;
  const int n = size * size;       // lanes per sub-block: 16 or 64
  const int per_wave = 64 / n;     // 4 or 1
  const int sub = lane / n, idx = lane % n;
  const int bx_n = w / size, n_blocks = bx_n * (h / size);
  const int shift = 2 * (bd - 8);
  const long long c1 =
      (long long)(((unsigned long long)(n * n) * 26634ull >> 12) << shift);
  const long long c2 =
      (long long)(((unsigned long long)(n * n) * 239708ull >> 12) << shift);
  const long long c4 = 255ll * 255ll;
  const int z = qp_raw;
  int wtmp = (int)((4 * z - 0.054 * z * z - 70) * strength);
  const int wq = (wtmp > 0 ? wtmp : 0) >> 4;
  const int w1 = 64 - (wq >> 1);
  const int w2 = 2 * wq;
  uint64_t total = 0;
  for (int b0 = 0; b0 < n_blocks; b0 += per_wave) {
    const int blk = b0 + sub;
    const bool active = blk < n_blocks;
    int o = 0, r = 0;
    if (active) {
      const int x = (blk % bx_n) * size + (idx % size);
      const int y = (blk / bx_n) * size + (idx / size);
      o = ld_s(a, (ptrdiff_t)y * sa + x);
      r = (int)b[(ptrdiff_t)y * sb + x];
    }
    const int d = o - r;
    int s_o = o, s_r = r;
    long long s_oo = o * o, s_rr = r * r, s_or = o * r, s_dd = d * d;
    if (n == 16) {
      s_o = group_sum<16>(s_o);
      s_r = group_sum<16>(s_r);
      s_oo = group_sum<16>(s_oo);
      s_rr = group_sum<16>(s_rr);
      s_or = group_sum<16>(s_or);
      s_dd = group_sum<16>(s_dd);
    } else {
      s_o = group_sum<64>(s_o);
      s_r = group_sum<64>(s_r);
      s_oo = group_sum<64>(s_oo);
      s_rr = group_sum<64>(s_rr);
      s_or = group_sum<64>(s_or);
      s_dd = group_sum<64>(s_dd);
    }
    if (active && idx == 0) {
      const long long orig_sum = s_o, reco_sum = s_r, oo = s_oo, rr = s_rr,
                      orr = s_or;
      long long ssd = s_dd;
      const double m = (1.0 * orig_sum - reco_sum) / n;
      const double aa = (c4 - m * m + c1) / (c4 + c1);
      const double bb =
          (2.0 * n * orr - 2 * orig_sum * reco_sum + c2) /
          (n * oo - orig_sum * orig_sum + n * rr - reco_sum * reco_sum + c2);
      ssd >>= shift;
      total += (uint64_t)(w1 * ssd +
                          w2 * (c4 >> ((8 - size) >> 1)) * (1 - aa * bb)) >>
               6;
    }
  }
  return group_sum<64>(total);
}

// ---- Compare() dispatch -----------------------------------------------------
template <typename T1>
__device__ __forceinline__ uint64_t wave_compare(int metric, int bd, int qp_raw,
                                                 int strength, int w, int h,
                                                 const T1 *a, int sa,
                                                 const uint16_t *b, int sb) {
  switch (metric) {
    case XVC_METRIC_SSD:
      return wave_ssd(w, h, a, sa, b, sb) >> (2 * (bd - 8));
    case XVC_METRIC_SATD:
      return wave_satd(bd, w, h, 0, a, sa, b, sb);
    case XVC_METRIC_SATD_ACONLY:
      return wave_satd(bd, w, h, wave_mean_diff(0, w, h, a, sa, b, sb), a, sa,
                       b, sb);
    case XVC_METRIC_SAD:
      return (uint64_t)(int64_t)wave_sad(w, h, 1, a, sa, b, sb) >> (bd - 8);
    case XVC_METRIC_SAD_FAST:
      return ((uint64_t)(int64_t)wave_sad(w, h / 2, 2, a, sa, b, sb) * 2) >>
             (bd - 8);
    case XVC_METRIC_SAD_ACONLY:
      return wave_sad_ac(0, bd, w, h, a, sa, b, sb);
    case XVC_METRIC_SAD_ACONLY_FAST:
      return wave_sad_ac(1, bd, w, h, a, sa, b, sb);
    case XVC_METRIC_STRUCTURAL_SSD:
      return wave_structural_ssd(bd, qp_raw, strength, w, h, a, sa, b, sb);
    default:
      return ~0ull;
  }
}

// grid: ceil(n/4); block: 256 = 4 waves, one candidate per wave.
__global__ void __launch_bounds__(256)
metric_batch_kernel(PlaneView pa, PlaneView pb, int bd, double weight,
                    int strength, const xvcgpu_metric_cand *cands, int n,
                    uint64_t *out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n) return;
  const xvcgpu_metric_cand cd = cands[c];
  const uint16_t *a = pa.p + (ptrdiff_t)cd.y * pa.stride + cd.x;
  const uint16_t *b =
      pb.p + (ptrdiff_t)(cd.y + cd.mv_y) * pb.stride + cd.x + cd.mv_x;
  const uint64_t dist = wave_compare(cd.metric, bd, cd.qp, strength, cd.w, cd.h,
                                     a, pa.stride, b, pb.stride);
  if ((threadIdx.x & 63) == 0) out[c] = (uint64_t)((double)dist * weight);
}

#endif  // XVCGPU_K_METRIC_H_
