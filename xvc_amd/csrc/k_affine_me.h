// k_affine_me.h -- T5: InterSearch::MotionEstAffine (inter_search.cc:664-749;
// uni-pred and the bi-pred refinement search of SearchBiIterative :394-435)
// with AffineGradientSearch (:751-851) and DeriveMvAffine
// (inter_prediction.cc:615-630), the whole iteration on the device.  One wave
// per CU.
//
// The reference keeps the gradients in float and the 4x5 normal equations in
// double, summed in raster order.  Every term is a multiple of 1/64 (1/8 for
// the right-hand side) and the totals stay below 2^53 for 8..12-bit samples
// and blocks up to 64x64 (|8c| < 2^21, 4096 samples), so each of those
// double sums is exact and therefore independent of the order: the lanes
// accumulate the same sums as 64-bit integers and the matrix handed to the
// elimination is bit-identical.  The elimination itself is a fixed sequence
// of IEEE double operations (no contraction), evaluated the same on every lane.
#ifndef XVCGPU_K_AFFINE_ME_H_
#define XVCGPU_K_AFFINE_ME_H_

#include "k_bipred.h"
#include "k_me.h"
#include "k_subpel.h"

// NW = h / 8 waves per CU (round 6; h / 16 before: a 16-high CU was ONE wave, 60 - 73 us,
// the longest launch of an engine round).  A wave owns a slab of the block for the
// plain-MC case, the distortions and the gradient sums: 8 rows where the SATD tiles are 8
// rows high (w >= h: 8x8 / 16x8 tiles; all NW waves), 16 rows for the tall blocks (w < h:
// 8x16 tiles; the first h / 16 waves, the others only share the affine MC of all
// sub-blocks, which is dealt over all threads).  Every sum is of integers, so the split
// changes no result.  The vectors, costs and the elimination are carried redundantly
// (and identically) by all lanes.
__device__ __forceinline__ int affine_slab_rows(int w, int h) { return w >= h ? 8 : 16; }

template <int NW>
struct __attribute__((aligned(16))) AffineMeShared {
  // filter intermediates: per wave a slab (plain MC: w x (rows + 7)), or for all
  // sub-blocks of the CU together (w x h x (sbh + 7) / sbh <= 2.75 w h)
  int16_t tmp[NW * 8 * 64 * 11 / 4];
  int sbmv[NW * 64][2];          // clipped vector of each (>= 4x4) sub-block
  uint16_t pred[NW * 8 * 64];    // the CU's prediction, row stride w
  int16_t target[NW * 8 * 64];   // bi-pred: 2 * orig - the other list's prediction
  long long part[NW][16];   // per-wave partial sums of the normal equations
  unsigned long long dpart[NW];
};

// InterPrediction::MotionCompAffine (inter_prediction.cc:1044-1136), luma, into
// s.pred.
template <int NW>
__device__ __forceinline__ void affine_me_mc(int bd, int bx, int by, int w, int h,
                                             const PlaneView &pr, const int mvin[3][2],
                                             AffineMeShared<NW> &s) {
  const int wave = threadIdx.x >> 6;
  int mv[3][2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    mv[i][0] = mvin[i][0];
    mv[i][1] = mvin[i][1];
    d_clip_mv(bx, by, pr.w, pr.h, mv[i][0], mv[i][1]);
  }
  __syncthreads();  // earlier readers of s.pred are done
  if (mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) {
    // plain MC: the wave's slab as a block of its own
    const int rows = affine_slab_rows(w, h);
    if (wave * rows < h) {
      const uint16_t *r = pr.p + (ptrdiff_t)(by + wave * rows + (mv[0][1] >> 4)) * pr.stride + bx +
                          (mv[0][0] >> 4);
      wave_interp_block<false>(bd, w, rows, mv[0][0] & 15, mv[0][1] & 15, r, pr.stride,
                               s.tmp + wave * (rows * 64 * 11 / 4), s.pred + wave * rows * w);
    }
    __syncthreads();
    return;
  }
  // All sub-blocks at once (they are 4x4 ... 16x16: one at a time would leave
  // most of a wave idle): the arithmetic per sample is MotionCompUniPred's
  // (wave_interp_block), with the sub-block's own phase.
  const int sbw = d_affine_subblock(mv[0][0], mv[0][1], mv[1][0], mv[1][1], w, 0);
  const int sbh = d_affine_subblock(mv[0][0], mv[0][1], mv[2][0], mv[2][1], h, 0);
  const int nsx = w / sbw, nsy = h / sbh, n_sub = nsx * nsy;
  const int lsw = 31 - __clz(sbw), lsh = 31 - __clz(sbh), lnx = 31 - __clz(nsx);
  const int T = 64 * NW, tid = threadIdx.x;
  {
    const int mv_max_x = (pr.w - bx + 8 - 1) * 16, mv_min_x = (-64 - bx - 8 + 1) * 16;
    const int mv_max_y = (pr.h - by + 8 - 1) * 16, mv_min_y = (-64 - by - 8 + 1) * 16;
    const int dhx = ((mv[1][0] - mv[0][0]) * 256) / w;
    const int dhy = ((mv[1][1] - mv[0][1]) * 256) / w;
    const int dvx = -dhy, dvy = dhx;
    for (int k = tid; k < n_sub; k += T) {
      const int iy = k >> lnx, ix = k & (nsx - 1);
      const int hor_x = mv[0][0] * 256 + dvx * sbh * iy + dhx * sbw * ix;
      const int hor_y = mv[0][1] * 256 + dvy * sbh * iy + dhy * sbw * ix;
      const int mx = (hor_x + dhx * (sbw >> 1) + dvx * (sbh >> 1)) >> 8;
      const int my = (hor_y + dhy * (sbw >> 1) + dvy * (sbh >> 1)) >> 8;
      s.sbmv[k][0] = d_clip3(mx, mv_min_x, mv_max_x);
      s.sbmv[k][1] = d_clip3(my, mv_min_y, mv_max_y);
    }
  }
  __syncthreads();
  const int rows = sbh + 7;  // intermediate rows of a sub-block
  {
    // horizontal pass of the sub-blocks with a fractional phase in both
    // directions: the 8 samples in one (2-byte aligned) 16-byte load, the taps
    // as packed pairs, four dot2
    const int shift = 6 - (14 - bd), offset = -(8192 << shift);
    for (int j = tid; j < rows * n_sub * sbw; j += T) {
      const int c = j & (sbw - 1), k = (j >> lsw) & (n_sub - 1), r = j >> (lsw + lnx + (31 - __clz(nsy)));
      const int mx = s.sbmv[k][0], my = s.sbmv[k][1];
      if (!(mx & 15) || !(my & 15)) continue;
      const int sx = (k & (nsx - 1)) << lsw, sy = (k >> lnx) << lsh;
      const uint16_t *p = pr.p + (ptrdiff_t)(by + sy + (my >> 4) + r - 3) * pr.stride + bx + sx +
                          (mx >> 4) + c - 3;
      const U16x8 v = *reinterpret_cast<const U16x8 *>(p);
      const uint4 t = *reinterpret_cast<const uint4 *>(kLumaTaps[mx & 15]);
      int sum = sp_dot2(v.v[0], t.x, 0);  // samples < 2^15: exact as signed pairs
      sum = sp_dot2(v.v[1], t.y, sum);
      sum = sp_dot2(v.v[2], t.z, sum);
      sum = sp_dot2(v.v[3], t.w, sum);
      s.tmp[(k * rows + r) * sbw + c] = (int16_t)((sum + offset) >> shift);
    }
  }
  __syncthreads();
  {
    const int smax = (1 << bd) - 1, lw = 31 - __clz(w);
    const int shift2 = 6 + (14 - bd), offset2 = (8192 << 6) + (1 << (shift2 - 1));
    for (int i = tid; i < w * h; i += T) {
      const int x = i & (w - 1), y = i >> lw;
      const int k = ((y >> lsh) << lnx) + (x >> lsw), c = x & (sbw - 1), r = y & (sbh - 1);
      const int mx = s.sbmv[k][0], my = s.sbmv[k][1];
      const int fx = mx & 15, fy = my & 15;
      const uint16_t *p = pr.p + (ptrdiff_t)(by + y + (my >> 4)) * pr.stride + bx + x + (mx >> 4);
      int v;
      if (fx == 0 && fy == 0) {
        v = p[0];
      } else if (fy == 0) {
        const int16_t *fh = kLumaTaps[fx];
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += (int)p[t - 3] * fh[t];
        v = d_clip_bd((sum + 32) >> 6, smax);
      } else if (fx == 0) {
        const int16_t *fv = kLumaTaps[fy];
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += (int)p[(ptrdiff_t)(t - 3) * pr.stride] * fv[t];
        v = d_clip_bd((int16_t)((sum + 32) >> 6), smax);
      } else {
        const int16_t *fv = kLumaTaps[fy];
        const int16_t *q = s.tmp + (k * rows + r) * sbw + c;
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += (int)q[t * sbw] * fv[t];
        v = d_clip_bd((int16_t)((sum + offset2) >> shift2), smax);
      }
      s.pred[i] = (uint16_t)v;
    }
  }
  __syncthreads();
}

// SampleMetric::CompareSample with kSad / kSatd over the whole block: the
// waves' slab parts summed, then the bit-depth normalisation of Compare().
template <int NW, typename TOrig>
__device__ __forceinline__ uint64_t affine_me_dist(int metric, int bd, int w, int h,
                                                   const TOrig *o, int os,
                                                   AffineMeShared<NW> &s) {
  const int wave = threadIdx.x >> 6, rows = affine_slab_rows(w, h);
  const TOrig *oo = o + (ptrdiff_t)wave * rows * os;
  const uint16_t *pp = s.pred + wave * rows * w;
  uint64_t part = 0;
  if (wave * rows >= h)
    ;  // (a tall block's upper waves: no slab)
  else if (metric == XVC_METRIC_SAD)
    part = (uint64_t)(int64_t)wave_sad(w, rows, 1, oo, os, pp, w);
  else if (w == h)
    part = wave_satd_tiles<8, 8>(w, 8, 0, oo, os, pp, w);
  else if (w > h)
    part = wave_satd_tiles<16, 8>(w, 8, 0, oo, os, pp, w);
  else
    part = wave_satd_tiles<8, 16>(w, 16, 0, oo, os, pp, w);
  s.dpart[wave] = part;
  __syncthreads();
  uint64_t total = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) total += s.dpart[k];
  __syncthreads();
  return total >> (bd - 8);
}

// ::lround into MvDelta's int members as the reference's x86-64 build does it:
// out of range gives the "integer indefinite" value whose low 32 bits are 0.
__device__ __forceinline__ int affine_lround(double v) {
  if (!(fabs(v) < 9.2e18)) return 0;
  return (int)lround(v);
}

// The elimination, back substitution and rounding of AffineGradientSearch
// (inter_search.cc:805-850) on the exact sums: S = 64 * matrix[r][c] for
// r <= c in the order 00 01 02 03 11 12 13 22 23 33, R = 8 * matrix[r][4].
// Lane 5 r + c of the wave holds matrix[r][c]: the updates of one pivot step
// are independent of each other (each element sees the reference's own
// multiply, divide, subtract, in that order), so a step costs one division
// instead of twelve; the back substitution is serial and carried by all lanes.
__device__ __forceinline__ double affine_lane_f64(double v, int src_lane) {
  return __shfl(v, src_lane, XVC_WAVE);
}

__device__ __forceinline__ int affine_bitrev4(int v) {
  return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3);
}

// `held`: the lane's share of the 14 totals after affine_reduce16 (lane & 15 =
// bit-reversed index); with several waves the per-wave totals are in `part`.
template <int NW>
__device__ __noinline__ void affine_solve(long long held, const long long (*part)[16],
                                          int width, int mvd[4]) {
#pragma clang fp contract(off)
  const int lane = ME2_LANE;
  const int r = lane < 20 ? lane / 5 : 0, c = lane < 20 ? lane - 5 * (lane / 5) : 0;
  double m;
  {
    // index of matrix[r][c] among the totals: S at (min, max) with row starts
    // 0, 4, 7, 9; the right-hand side at 10 + r
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int start = lo == 0 ? 0 : (lo == 1 ? 4 : (lo == 2 ? 7 : 9));
    const int kk = c < 4 ? start + hi - lo : 10 + r;
    long long val;
    if (NW == 1) {
      val = __shfl(held, affine_bitrev4(kk), XVC_WAVE);
    } else {
      val = 0;
#pragma unroll
      for (int v = 0; v < NW; v++) val += part[v][kk];
    }
    m = c < 4 ? (double)val / 64.0 : (double)val / 8.0;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int best = i;
    double best_val = fabs(affine_lane_f64(m, 5 * i + i));
#pragma unroll
    for (int j = i + 1; j < 4; j++) {
      const double v = fabs(affine_lane_f64(m, 5 * j + i));
      if (v > best_val) {
        best = j;
        best_val = v;
      }
    }
    // swap rows i and best (uniform decision)
    {
      const int other = r == i ? best : (r == best ? i : r);
      m = affine_lane_f64(m, 5 * other + c);
    }
    const double pivot = affine_lane_f64(m, 5 * i + i);
    const double mik = affine_lane_f64(m, 5 * i + c);
    const double mji = affine_lane_f64(m, 5 * r + i);
    if (r > i && c > i && pivot != 0.0) {
      const double prod = mik * mji;
      const double q = prod / pivot;
      m = m - q;
    }
  }
  // back substitution, identically on every lane
  double u[4][5];
#pragma unroll
  for (int rr = 0; rr < 4; rr++)
#pragma unroll
    for (int cc = rr; cc < 5; cc++) u[rr][cc] = affine_lane_f64(m, 5 * rr + cc);
  double params[4] = {0, 0, 0, 0};
  if (u[3][3] != 0.0) params[3] = u[3][4] / u[3][3];
#pragma unroll
  for (int row = 2; row >= 0; row--) {
    double sum = 0;
#pragma unroll
    for (int col = row + 1; col < 4; col++) {
      const double prod = u[row][col] * params[col];
      sum = sum + prod;
    }
    if (u[row][row] != 0.0) params[row] = (u[row][4] - sum) / u[row][row];
  }
  const double p1w = params[1] * (double)width;
  const double p3w = -params[3] * (double)width;
  mvd[0] = affine_lround(4.0 * params[0]);
  mvd[1] = affine_lround(4.0 * params[2]);
  mvd[2] = affine_lround(4.0 * (p1w + params[0]));
  mvd[3] = affine_lround(4.0 * (p3w + params[2]));
}

// AffineGradientSearch on the prediction in s.pred and err = orig - pred.
template <int NW, typename TOrig>
__device__ __forceinline__ void affine_gradient_search(int w, int h, const TOrig *o,
                                                       int os, AffineMeShared<NW> &s,
                                                       int mvd[4]) {
  const int lane = ME2_LANE, wave = threadIdx.x >> 6, lw = 31 - __clz(w);
  long long S[10], R[4];
#pragma unroll
  for (int k = 0; k < 10; k++) S[k] = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) R[k] = 0;
  const int rows = affine_slab_rows(w, h);
  const int i_end = (wave + 1) * rows <= h ? (wave + 1) * rows * w : 0;   // (no slab: empty)
  for (int i = wave * rows * w + lane; i < i_end; i += 64) {
    const int x = i & (w - 1), y = i >> lw;
    // border gradients are copies of the nearest interior one (:772-783)
    const int xc = d_clip3(x, 1, w - 2), yc = d_clip3(y, 1, h - 2);
    const uint16_t *p = s.pred + yc * w + xc;
    const int a0 = p[-w - 1], a1 = p[-w], a2 = p[-w + 1];
    const int b0 = p[-1], b2 = p[1];
    const int c0 = p[w - 1], c1 = p[w], c2 = p[w + 1];
    const int kh = -a0 + a2 - 2 * b0 + 2 * b2 - c0 + c2;  // 8 * affine_delta_hor_
    const int kv = -a0 - 2 * a1 - a2 + c0 + 2 * c1 + c2;  // 8 * affine_delta_ver_
    // |c| < 2^21: 32 x 32 -> 64-bit multiply-adds
    const int c[4] = {kh, x * kh + y * kv, kv, y * kh - x * kv};
    const int e = (int)o[(ptrdiff_t)y * os + x] - (int)s.pred[i];
    int k = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int cc = r; cc < 4; cc++, k++) S[k] += (long long)c[r] * c[cc];
      R[r] += (long long)e * c[r];
    }
  }
  // reduce-scatter butterfly: 17 shuffles instead of 84 - after it the lane
  // whose low four bits are the bit-reversed index k holds total k
  long long v[16];
#pragma unroll
  for (int k = 0; k < 10; k++) v[k] = S[k];
#pragma unroll
  for (int k = 0; k < 4; k++) v[10 + k] = R[k];
  v[14] = v[15] = 0;
#pragma unroll
  for (int half = 8, bit = 1; half >= 1; half >>= 1, bit <<= 1) {
    const bool upper = (lane & bit) != 0;
#pragma unroll
    for (int j = 0; j < half; j++) {
      const long long keep = upper ? v[j + half] : v[j];
      const long long send = upper ? v[j] : v[j + half];
      v[j] = keep + __shfl_xor(send, bit, XVC_WAVE);
    }
  }
  long long held = v[0];
  held += __shfl_xor(held, 16, XVC_WAVE);
  held += __shfl_xor(held, 32, XVC_WAVE);
  if (NW > 1) {
    if (lane < 16) s.part[wave][affine_bitrev4(lane)] = held;
    __syncthreads();
  }
  affine_solve<NW>(held, s.part, w, mvd);
}

// The search proper on a target of samples (uni) or residuals (bi).
template <int NW, typename TOrig>
__device__ __forceinline__ void affine_me_search(int bd, const xvcgpu_affine_me_block &b,
                                                 const PlaneView &ref, const TOrig *o, int os,
                                                 AffineMeShared<NW> &s,
                                                 xvcgpu_affine_me_result *out) {
  const int w = b.w, h = b.h, bx = b.x, by = b.y;
  const uint32_t lambda = b.lambda16;
  const bool bipred = (b.flags & XVC_AFFINE_ME_BIPRED) != 0;
  const int bi_shift = bipred ? 1 : 0, max_iterations = bipred ? 5 : 7;
  int mvp[3][2], best_mv[3][2], mv[3][2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    mvp[i][0] = best_mv[i][0] = b.mvp[i][0];
    mvp[i][1] = best_mv[i][1] = b.mvp[i][1];
  }
  auto bits3 = [&](const int v[3][2]) -> uint32_t {
    return d_mvd_bits(mvp[0][0], mvp[0][1], v[0][0], v[0][1], 0) +
           d_mvd_bits(mvp[1][0], mvp[1][1], v[1][0], v[1][1], 0);
  };
  auto dist_of = [&](int metric) -> uint64_t {
    return affine_me_dist(metric, bd, w, h, o, os, s) >> bi_shift;
  };
  affine_me_mc(bd, bx, by, w, h, ref, mvp, s);
  uint64_t best_dist = dist_of(XVC_METRIC_SAD);
  uint64_t best_cost = best_dist + ((uint32_t)(lambda * bits3(best_mv)) >> 16);
  if (b.flags & XVC_AFFINE_ME_HAS_BOOTSTRAP) {
    int boot[3][2];
    bool same = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      boot[i][0] = b.bootstrap[i][0];
      boot[i][1] = b.bootstrap[i][1];
      same = same && boot[i][0] == mvp[i][0] && boot[i][1] == mvp[i][1];
    }
    if (!same) {
      affine_me_mc(bd, bx, by, w, h, ref, boot, s);
      const uint64_t dist = dist_of(XVC_METRIC_SAD);
      const uint64_t cost = dist + ((uint32_t)(lambda * bits3(boot)) >> 16);
      if (cost < best_cost || bipred) {  // force_mv_bootstrap
#pragma unroll
        for (int i = 0; i < 3; i++) {
          best_mv[i][0] = boot[i][0];
          best_mv[i][1] = boot[i][1];
        }
      } else {
        affine_me_mc(bd, bx, by, w, h, ref, best_mv, s);
      }
    }
  }
  best_dist = dist_of(XVC_METRIC_SATD);
  best_cost = best_dist + ((uint32_t)(lambda * bits3(best_mv)) >> 16);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    mv[i][0] = best_mv[i][0];
    mv[i][1] = best_mv[i][1];
  }
  uint32_t iterations = 0;
  for (int iter = 0; iter < max_iterations; iter++) {
    int mvd[4];
    affine_gradient_search(w, h, o, os, s, mvd);
    if (!(mvd[0] | mvd[1] | mvd[2] | mvd[3])) break;
    iterations++;
    mv[0][0] += mvd[0] * 4;  // MotionVector += MvDelta (1/4 -> 1/16 pel)
    mv[0][1] += mvd[1] * 4;
    mv[1][0] += mvd[2] * 4;
    mv[1][1] += mvd[3] * 4;
    // DeriveMvAffine
    d_clip_mv(bx, by, ref.w, ref.h, mv[0][0], mv[0][1]);
    d_clip_mv(bx, by, ref.w, ref.h, mv[1][0], mv[1][1]);
    mv[2][0] = mv[0][0] - (mv[1][1] - mv[0][1]) * h / w;
    mv[2][1] = mv[0][1] + (mv[1][0] - mv[0][0]) * h / w;
    d_clip_mv(bx, by, ref.w, ref.h, mv[2][0], mv[2][1]);
    affine_me_mc(bd, bx, by, w, h, ref, mv, s);
    const uint64_t dist = dist_of(XVC_METRIC_SATD);
    const uint64_t cost = dist + ((uint32_t)(lambda * bits3(mv)) >> 16);
    if (cost < best_cost) {
      best_cost = cost;
      best_dist = dist;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        best_mv[i][0] = mv[i][0];
        best_mv[i][1] = mv[i][1];
      }
    }
  }
  if (threadIdx.x == 0) {
    xvcgpu_affine_me_result r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      r.mv[i][0] = best_mv[i][0];
      r.mv[i][1] = best_mv[i][1];
    }
    r.dist = (uint32_t)best_dist;
    r.iterations = iterations;
    *out = r;
  }
}

// grid: n CUs; block: 64 * NW.  The instance with NW waves takes the jobs
// whose height is 8 * NW and leaves the others to its siblings.
template <int NW>
__device__ __forceinline__ void
affine_me_body(const PlaneView &orig, const PlaneView &ref_arg, const PlaneView &ref_other_arg,
               int bd, const xvcgpu_affine_me_block *blocks, int n, xvcgpu_affine_me_result *out,
               const RefTable *refs = nullptr, const uint8_t *slots = nullptr) {
  __shared__ AffineMeShared<NW> s;
  const int bi = blockIdx.x;
  if (bi >= n) return;
  // (the *_refs form: slots[2 * job] = the searched picture, [2 * job + 1] = the other list's)
  int slot_s = 0, slot_o = 0;
  if (slots) {
    slot_s = __builtin_amdgcn_readfirstlane((int)slots[2 * bi]);
    slot_o = __builtin_amdgcn_readfirstlane((int)slots[2 * bi + 1]);
    if (slot_s >= refs->n) return;
    if (slot_o >= refs->n) slot_o = slot_s;   // (uni-prediction: not read)
  }
  const PlaneView ref = slots ? refs->pic[slot_s].c[0] : ref_arg;
  const PlaneView ref_other = slots ? refs->pic[slot_o].c[0] : ref_other_arg;
  const xvcgpu_affine_me_block b = blocks[bi];
  {
    // CodingUnit::CanUseAffine: width, height > 8 (coding_unit.h:308): 16, 32, 64
    const bool valid = (b.w == 16 || b.w == 32 || b.w == 64) &&
                       (b.h == 16 || b.h == 32 || b.h == 64);
    if (NW == 2 && !valid) {  // nobody takes it: the XVCGPU_AFFINE_ME_UNSUPPORTED record
      if (threadIdx.x == 0) {
        xvcgpu_affine_me_result r;
        for (int i = 0; i < 3; i++) r.mv[i][0] = r.mv[i][1] = 0;
        r.dist = 0xffffffffu;
        r.iterations = 0xffffffffu;
        out[bi] = r;
      }
      return;
    }
    if (!valid) return;
  }
  if (b.h != 8 * NW) return;
  const uint16_t *o = orig.p + (ptrdiff_t)b.y * orig.stride + b.x;
  if (b.flags & XVC_AFFINE_ME_BIPRED) {
    // SearchBiIterative :415-420: the other list's prediction, SubtractWeighted
    int other[3][2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      other[i][0] = b.other_mv[i][0];
      other[i][1] = b.other_mv[i][1];
    }
    affine_me_mc(bd, b.x, b.y, b.w, b.h, ref_other, other, s);
    const int w = b.w, lw = 31 - __clz(w);
    for (int i = threadIdx.x; i < w * b.h; i += 64 * NW)
      s.target[i] = (int16_t)(2 * (int)o[(ptrdiff_t)(i >> lw) * orig.stride + (i & (w - 1))] -
                              (int)s.pred[i]);
    __syncthreads();
    affine_me_search<NW, int16_t>(bd, b, ref, s.target, w, s, out + bi);
  } else {
    affine_me_search<NW, uint16_t>(bd, b, ref, o, (int)orig.stride, s, out + bi);
  }
}

template <int NW>
__global__ void __launch_bounds__(64 * NW)
affine_me_kernel(PlaneView orig, PlaneView ref, PlaneView ref_other, int bd,
                 const xvcgpu_affine_me_block *blocks, int n, xvcgpu_affine_me_result *out) {
  affine_me_body<NW>(orig, ref, ref_other, bd, blocks, n, out);
}

// The affine searches of one CU state into several reference pictures in one launch
// (xvcgpu_affine_me_batch_refs).
template <int NW>
__global__ void __launch_bounds__(64 * NW)
affine_me_refs_kernel(PlaneView orig, RefTable refs, const uint8_t *slots, int bd,
                      const xvcgpu_affine_me_block *blocks, int n,
                      xvcgpu_affine_me_result *out) {
  affine_me_body<NW>(orig, orig, orig, bd, blocks, n, out, &refs, slots);
}

#endif  // XVCGPU_K_AFFINE_ME_H_
