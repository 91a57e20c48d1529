// k_subpel.h -- T3 fast path: the sub-pel SATD sweep for blocks with both
// sides >= 8 (8x8, 16x8 or 8x16 SATD tiles) at bit depth <= 10, on packed
// 16-bit math.
// Same arithmetic as me2_build_hplanes / me2_satd_cands in k_me2.h
// (inter_prediction.cc:1207-1448, sample_metric.cc:316-641), fewer VALU
// instructions:
//   * the filtered planes are kept COLUMN-major in LDS, so the vertical 8-tap
//     filter of 8 outputs of one column reads 16 consecutive int16 (two
//     ds_read_b128) and is 5 v_dot2c_i32_i16 per output: rows are consumed as
//     aligned pairs, an odd start uses the tap set shifted by one
//     ((0,t0),(t1,t2),(t3,t4),(t5,t6),(t7,0));
//   * lane = one column of one 8x8 tile for TWO candidates, one in each
//     16-bit half: clip, residual and the Hadamard butterflies are v_pk_*;
//     the vertical WHT is in registers, the horizontal one goes through
//     ds_swizzle (LDS crossbar, no VALU);
//   * the last butterfly stage is never formed: |a+b| + |a-b| = 2 max(|a|,|b|).
// 16-bit safety: |orig - pred| <= 1023 (bd <= 10); after k stages <= 1023*2^k;
// five stages = 32736 < 2^15; the sum of two maxima 65472 < 2^16.
#ifndef XVCGPU_K_SUBPEL_H_
#define XVCGPU_K_SUBPEL_H_

#include "dev_common.h"
#include "dev_tables.h"

typedef short sp_v2s __attribute__((ext_vector_type(2)));
typedef unsigned short sp_v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ sp_v2s sp_s2(uint32_t v) { return __builtin_bit_cast(sp_v2s, v); }
__device__ __forceinline__ uint32_t sp_u(sp_v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ int sp_dot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(sp_s2(a), sp_s2(b), c, false);
}
__device__ __forceinline__ uint32_t sp_pack_taps(int lo, int hi) {
  return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}
// d = a * b + c on both 16-bit halves
__device__ __forceinline__ uint32_t sp_pk_mad(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int PATTERN>
__device__ __forceinline__ uint32_t sp_swizzle(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, PATTERN);
}

// Per-candidate parameters of the vertical stage, 16 dwords.
struct __attribute__((aligned(16))) SpCand {
  uint32_t e[5];   // tap pairs for even outputs of the column
  uint32_t o[5];   // tap pairs for odd outputs
  int off, sh;     // rounding offset, shift
  int plane;       // int16 index of the plane's (x = 0, stored row 0)
  int pad[3];
};

// Column-major plane of one x-phase: p14 = FilterHorSampleShort (14 bit), or
// the unfiltered samples when fx == 0; column
// stride h + 8, stored row r <-> picture row r - 4 relative to the full-pel
// position.  win: row-major window, cols -8..w+7 (row stride w + 16).
__device__ __forceinline__ void sp_build_planes(const uint16_t *win, int16_t *p14,
                                                const int16_t (*taps)[8],
                                                int bd, int w, int h, int pel_x, int fx,
                                                int tid = threadIdx.x & 63, int nthr = 64) {
  const int lane = tid;   // tid / nthr: this thread's place in the wave (or team of waves)
  const int ws = w + 16, rs = h + 8;
  const int hw = w >> 1, lhw = 31 - __clz(hw);
  const int n = rs * hw;  // units: (row, pair of columns)
  if (fx == 0) {
    for (int i = lane; i < n; i += nthr) {
      const int r = i >> lhw, x0 = (i & (hw - 1)) << 1;
      const uint16_t *src = win + r * ws + x0 + pel_x + 8;
      p14[x0 * rs + r] = (int16_t)src[0];
      p14[(x0 + 1) * rs + r] = (int16_t)src[1];
    }
    return;
  }
  const int16_t *f = taps[fx];  // LDS copy of kLumaTaps
  const uint32_t a0 = sp_pack_taps(f[0], f[1]), a1 = sp_pack_taps(f[2], f[3]),
                 a2 = sp_pack_taps(f[4], f[5]), a3 = sp_pack_taps(f[6], f[7]);
  const uint32_t b0 = sp_pack_taps(0, f[0]), b1 = sp_pack_taps(f[1], f[2]),
                 b2 = sp_pack_taps(f[3], f[4]), b3 = sp_pack_taps(f[5], f[6]),
                 b4 = sp_pack_taps(f[7], 0);
  const int shift = 6 - (14 - bd), offset = -(8192 << shift);
  // first tap of output x0 sits at window column c0 = x0 + pel_x + 5
  const bool odd = ((pel_x + 5) & 1) != 0;
  const uint32_t *win32 = reinterpret_cast<const uint32_t *>(win);
  for (int i = lane; i < n; i += nthr) {
    const int r = i >> lhw, x0 = (i & (hw - 1)) << 1;
    const uint32_t *d = win32 + ((r * ws + x0 + pel_x + 5) >> 1);
    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
    int s0, s1;
    if (!odd) {
      s0 = sp_dot2(d0, a0, 0); s0 = sp_dot2(d1, a1, s0); s0 = sp_dot2(d2, a2, s0);
      s0 = sp_dot2(d3, a3, s0);
      s1 = sp_dot2(d0, b0, 0); s1 = sp_dot2(d1, b1, s1); s1 = sp_dot2(d2, b2, s1);
      s1 = sp_dot2(d3, b3, s1); s1 = sp_dot2(d4, b4, s1);
    } else {
      s0 = sp_dot2(d0, b0, 0); s0 = sp_dot2(d1, b1, s0); s0 = sp_dot2(d2, b2, s0);
      s0 = sp_dot2(d3, b3, s0); s0 = sp_dot2(d4, b4, s0);
      s1 = sp_dot2(d1, a0, 0); s1 = sp_dot2(d2, a1, s1); s1 = sp_dot2(d3, a2, s1);
      s1 = sp_dot2(d4, a3, s1);
    }
    p14[x0 * rs + r] = (int16_t)((s0 + offset) >> shift);
    p14[(x0 + 1) * rs + r] = (int16_t)((s1 + offset) >> shift);
  }
}

// Tap sets of a candidate: q = (stored row of its first tap for output row 0)
// in {0, 1}; fy = vertical phase (0 -> identity taps).
__device__ __forceinline__ void sp_fill_taps(SpCand &c, const int16_t (*taps)[8], int fy,
                                             int q) {
  const int16_t *t = taps[fy];
  const uint32_t a0 = sp_pack_taps(t[0], t[1]), a1 = sp_pack_taps(t[2], t[3]),
                 a2 = sp_pack_taps(t[4], t[5]), a3 = sp_pack_taps(t[6], t[7]);
  const uint32_t b0 = sp_pack_taps(0, t[0]), b1 = sp_pack_taps(t[1], t[2]),
                 b2 = sp_pack_taps(t[3], t[4]), b3 = sp_pack_taps(t[5], t[6]),
                 b4 = sp_pack_taps(t[7], 0);
  if (q == 0) {
    c.e[0] = a0; c.e[1] = a1; c.e[2] = a2; c.e[3] = a3; c.e[4] = 0;
    c.o[0] = b0; c.o[1] = b1; c.o[2] = b2; c.o[3] = b3; c.o[4] = b4;
  } else {
    c.e[0] = b0; c.e[1] = b1; c.e[2] = b2; c.e[3] = b3; c.e[4] = b4;
    c.o[0] = 0; c.o[1] = a0; c.o[2] = a1; c.o[3] = a2; c.o[4] = a3;
  }
}

// Vertical filter of one tile column (8 outputs) of one candidate: raw
// (acc >> sh) values, not yet narrowed / clipped.
__device__ __forceinline__ void sp_vfilter8(const int16_t *lds, const SpCand &c, int col_off,
                                            int out[8]) {
  const uint4 *src = reinterpret_cast<const uint4 *>(lds + c.plane + col_off);
  const uint4 lo = src[0], hi = src[1];
  const uint32_t p[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int m = 0; m < 4; m++) {
    int e = c.off, o = c.off;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      if (m + i < 8) {
        e = sp_dot2(p[m + i], c.e[i], e);
        o = sp_dot2(p[m + i], c.o[i], o);
      }
    }
    out[2 * m] = e >> c.sh;
    out[2 * m + 1] = o >> c.sh;
  }
}

// SATD of `ncand` candidates (params in cand[]) against origc (the original
// block, column-major, column stride h) with TW x TH tiles: 8x8 (square
// blocks), 16x8 (w > h) or 8x16 (w < h), ComputeSatdNxM's choice for blocks
// with both sides >= 8 (sample_metric.cc:403-641).  Adds the normalised tile
// sums into dist[c].  w, h multiples of TW, TH; bd <= 10.
//
// lane = one column of one tile for two candidates (one per 16-bit half); the
// TH rows of the column are registers.  The vertical butterflies (log2 TH
// stages) run in registers, the horizontal ones (log2 TW stages) across the TW
// lanes of the tile through ds_swizzle.  16-bit budget at bd 10: a residual
// is <= 1023, five formed stages reach 32736.  The 8x8 tile forms 3 + 2 stages
// and folds the sixth (|a+b| + |a-b| = 2 max(|a|,|b|)).  The 128-sample tiles
// have seven stages: five are formed, the sixth is taken in MAGNITUDE only -
// for u, v of the fifth stage |u+v| and |u-v| are |u|+|v| and ||u|-|v||
// (which is which follows from the signs), both < 2^16 unsigned - and the
// seventh is the max fold on those magnitudes.
template <int K>
__device__ __forceinline__ uint32_t sp_swz_xor(uint32_t v) {
  return sp_swizzle<(K << 10) | 0x1F>(v);
}
__device__ __forceinline__ uint32_t sp_pk_abs(uint32_t v) {
  const sp_v2s x = sp_s2(v);
  return sp_u(__builtin_elementwise_max(x, sp_s2(0u) - x));
}

template <int TW, int TH>
__device__ __forceinline__ void sp_satd_pairs_t(const int16_t *lds, const SpCand *cand,
                                                const uint16_t *origc, uint32_t *dist,
                                                int bd, int w, int h, int ncand, int tid,
                                                int nthr) {
  static_assert((TW == 8 && TH == 8) || (TW == 16 && TH == 8) || (TW == 8 && TH == 16), "tile");
  constexpr int LT = TW == 16 ? 4 : 3;          // stages across lanes
  constexpr int HF = TH == 16 ? 1 : 2;          // of which formed as values
  constexpr bool MAG = LT - HF == 2;            // one magnitude-only stage before the fold
  const int lane = threadIdx.x & 63;
  const int rs = h + 8;
  const int tiles_x = w / TW;
  const int upp = (w * h) / TH;                 // units (tile columns) per pair
  const int npairs = (ncand + 1) >> 1;
  const int total = upp * npairs;
  const uint32_t smax2 = (uint32_t)((1 << bd) - 1) * 0x10001u;
  for (int g0 = tid & ~63; g0 < total; g0 += nthr) {   // whole waves stay in step
    const int g = g0 + lane;
    const bool active = g < total;
    const int gg = active ? g : 0;
    const int pr = gg / upp, u = gg - pr * upp;
    const int tile = u / TW, col = u & (TW - 1);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x = tx * TW + col;
    const int ca = 2 * pr, cb = (2 * pr + 1 < ncand) ? 2 * pr + 1 : 2 * pr;
    const int col_off = x * rs + ty * TH;
    uint32_t m[TH];
#pragma unroll
    for (int r8 = 0; r8 < TH / 8; r8++) {
      int va[8], vb[8];
      sp_vfilter8(lds, cand[ca], col_off + 8 * r8, va);
      // an odd candidate count leaves the last sweep with one candidate in both
      // halves (the half-pel pass has nine): filter it once (wave-uniform test)
      if (__builtin_amdgcn_ballot_w64(active && cb != ca) != 0) {
        sp_vfilter8(lds, cand[cb], col_off + 8 * r8, vb);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) vb[j] = va[j];
      }
      // 8 originals of the column, each replicated into both halves
      const uint4 o4 = *reinterpret_cast<const uint4 *>(origc + x * h + ty * TH + 8 * r8);
      const uint32_t ow[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t ov = __builtin_amdgcn_perm(ow[j >> 1], ow[j >> 1],
                                                  (j & 1) ? 0x03020302u : 0x01000100u);
        // narrow to int16 (pack keeps the low halves), clip to [0, smax]
        sp_v2s pk;
        pk.x = (short)va[j];
        pk.y = (short)vb[j];
        pk = __builtin_elementwise_min(__builtin_elementwise_max(pk, sp_s2(0u)), sp_s2(smax2));
        m[8 * r8 + j] = sp_u(sp_s2(ov) - pk);
      }
    }
    // vertical WHT (down the column) in registers
#pragma unroll
    for (int len = 1; len < TH; len <<= 1)
#pragma unroll
      for (int i = 0; i < TH; i += len << 1)
#pragma unroll
        for (int j = i; j < i + len; j++) {
          const sp_v2s a = sp_s2(m[j]), b = sp_s2(m[j + len]);
          m[j] = sp_u(a + b);
          m[j + len] = sp_u(a - b);
        }
    // horizontal WHT across the TW lanes of the tile: formed stages
    {
      const uint32_t sg = (col & 1) ? 0xffffffffu : 0x00010001u;
#pragma unroll
      for (int j = 0; j < TH; j++) m[j] = sp_pk_mad(m[j], sg, sp_swz_xor<1>(m[j]));
    }
    if (HF == 2) {
      const uint32_t sg = (col & 2) ? 0xffffffffu : 0x00010001u;
#pragma unroll
      for (int j = 0; j < TH; j++) m[j] = sp_pk_mad(m[j], sg, sp_swz_xor<2>(m[j]));
    }
    uint32_t sa = 0, sb = 0;
    if (!MAG) {
      // last stage (xor 4) folded into the absolute sum: 2 * max(|a|, |b|),
      // counted once by each lane of the pair
      uint32_t mx[TH];
#pragma unroll
      for (int j = 0; j < TH; j++) {
        const sp_v2u av = __builtin_bit_cast(sp_v2u, sp_pk_abs(m[j]));
        const sp_v2u ot = __builtin_bit_cast(
            sp_v2u, sp_swz_xor<4>(__builtin_bit_cast(uint32_t, av)));
        mx[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(av, ot));
      }
#pragma unroll
      for (int j = 0; j < TH; j += 2) {
        const sp_v2u t2 = __builtin_bit_cast(sp_v2u, mx[j]) + __builtin_bit_cast(sp_v2u, mx[j + 1]);
        sa = __builtin_amdgcn_udot2(t2, (sp_v2u){1, 0}, sa, false);
        sb = __builtin_amdgcn_udot2(t2, (sp_v2u){0, 1}, sb, false);
      }
    } else {
      constexpr int KM = 1 << HF, KF = 2 << HF;   // magnitude stage, fold stage
      // this lane forms u + v (bit clear) or v - u (bit set) of the pair
      const uint32_t flip = (col & KM) ? 0xffffffffu : 0u;
#pragma unroll
      for (int j = 0; j < TH; j++) {
        const uint32_t uu = m[j], vv = sp_swz_xor<KM>(uu);
        const sp_v2u au = __builtin_bit_cast(sp_v2u, sp_pk_abs(uu));
        const sp_v2u av = __builtin_bit_cast(sp_v2u, sp_pk_abs(vv));
        const uint32_t sum = __builtin_bit_cast(uint32_t, au + av);
        const uint32_t dif = __builtin_bit_cast(
            uint32_t, __builtin_elementwise_max(au, av) - __builtin_elementwise_min(au, av));
        // halves whose signs differ (all-ones), swapped for the subtracting lane
        const sp_v2s sx = sp_s2(uu ^ vv) >> (sp_v2s){15, 15};
        const uint32_t sel = sp_u(sx) ^ flip;
        const uint32_t mag = (dif & sel) | (sum & ~sel);
        const sp_v2u mg = __builtin_bit_cast(sp_v2u, mag);
        const sp_v2u ot = __builtin_bit_cast(sp_v2u, sp_swz_xor<KF>(mag));
        const sp_v2u mx = __builtin_elementwise_max(mg, ot);
        sa = __builtin_amdgcn_udot2(mx, (sp_v2u){1, 0}, sa, false);
        sb = __builtin_amdgcn_udot2(mx, (sp_v2u){0, 1}, sb, false);
      }
    }
    // tile totals over the TW columns
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0xB1, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0xB1, 0xF, 0xF, false);
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0x4E, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0x4E, 0xF, 0xF, false);
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0x141, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0x141, 0xF, 0xF, false);
    if (TW == 16) {
      sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0x140, 0xF, 0xF, false);
      sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0x140, 0xF, 0xF, false);
    }
    if (active && col == 0) {
      if (TW == TH) {
        atomicAdd(&dist[ca], (sa + 2) >> 2);
        if (cb != ca) atomicAdd(&dist[cb], (sb + 2) >> 2);
      } else {
        const double nrm = sqrt((double)(TW * TH));
        atomicAdd(&dist[ca], (uint32_t)(int)(2.0 * (double)sa / nrm));
        if (cb != ca) atomicAdd(&dist[cb], (uint32_t)(int)(2.0 * (double)sb / nrm));
      }
    }
  }
}

__device__ __forceinline__ void sp_satd_pairs(const int16_t *lds, const SpCand *cand,
                                              const uint16_t *origc, uint32_t *dist,
                                              int bd, int w, int h, int ncand,
                                              int tid = threadIdx.x & 63, int nthr = 64) {
  if (w == h) sp_satd_pairs_t<8, 8>(lds, cand, origc, dist, bd, w, h, ncand, tid, nthr);
  else if (w > h) sp_satd_pairs_t<16, 8>(lds, cand, origc, dist, bd, w, h, ncand, tid, nthr);
  else sp_satd_pairs_t<8, 16>(lds, cand, origc, dist, bd, w, h, ncand, tid, nthr);
}

#endif  // XVCGPU_K_SUBPEL_H_
