// k_subpel.h -- T3 fast path: the sub-pel SATD sweep for square blocks with
// 8x8 SATD tiles (w == h >= 8) at bit depth <= 10, on packed 16-bit math.
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

// Column-major planes of one x-phase: p14 = FilterHorSampleShort (14 bit),
// ps = FilterHorSampleSample (or the unfiltered samples when fx == 0); column
// stride h + 8, stored row r <-> picture row r - 4 relative to the full-pel
// position.  win: row-major window, cols -8..w+7 (row stride w + 16).
__device__ __forceinline__ void sp_build_planes(const uint16_t *win, int16_t *p14,
                                                int16_t *ps, const int16_t (*taps)[8],
                                                int bd, int w, int h, int pel_x, int fx) {
  const int lane = threadIdx.x & 63;
  const int ws = w + 16, rs = h + 8;
  const int hw = w >> 1, lhw = 31 - __clz(hw);
  const int n = rs * hw;  // units: (row, pair of columns)
  if (fx == 0) {
    for (int i = lane; i < n; i += 64) {
      const int r = i >> lhw, x0 = (i & (hw - 1)) << 1;
      const uint16_t *src = win + r * ws + x0 + pel_x + 8;
      ps[x0 * rs + r] = (int16_t)src[0];
      ps[(x0 + 1) * rs + r] = (int16_t)src[1];
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
  const int smax = (1 << bd) - 1;
  // first tap of output x0 sits at window column c0 = x0 + pel_x + 5
  const bool odd = ((pel_x + 5) & 1) != 0;
  const uint32_t *win32 = reinterpret_cast<const uint32_t *>(win);
  for (int i = lane; i < n; i += 64) {
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
    ps[x0 * rs + r] = (int16_t)d_clip_bd((s0 + 32) >> 6, smax);
    ps[(x0 + 1) * rs + r] = (int16_t)d_clip_bd((s1 + 32) >> 6, smax);
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
// block, column-major, column stride h).  Adds the
// normalised 8x8 tile sums (ComputeSatdNxM with the square rule (s+2)>>2,
// sample_metric.cc:403-641) into dist[c].  w == h, multiple of 8; bd <= 10.
__device__ __forceinline__ void sp_satd_pairs(const int16_t *lds, const SpCand *cand,
                                              const uint16_t *origc, uint32_t *dist,
                                              int bd, int w, int h, int ncand) {
  const int lane = threadIdx.x & 63;
  const int rs = h + 8;
  const int tiles_x = w >> 3;
  const int upp = tiles_x * (h >> 3) * 8;  // units (tile columns) per pair
  const int npairs = (ncand + 1) >> 1;
  const int total = upp * npairs;
  const uint32_t smax2 = (uint32_t)((1 << bd) - 1) * 0x10001u;
  for (int g0 = 0; g0 < total; g0 += 64) {
    const int g = g0 + lane;
    const bool active = g < total;
    const int gg = active ? g : 0;
    const int pr = gg / upp, u = gg - pr * upp;
    const int tile = u >> 3, col = u & 7;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x = tx * 8 + col;
    const int ca = 2 * pr, cb = (2 * pr + 1 < ncand) ? 2 * pr + 1 : 2 * pr;
    const int col_off = x * rs + ty * 8;
    int va[8], vb[8];
    sp_vfilter8(lds, cand[ca], col_off, va);
    sp_vfilter8(lds, cand[cb], col_off, vb);
    // 8 originals of the column, each replicated into both halves
    const uint4 o4 = *reinterpret_cast<const uint4 *>(origc + x * h + ty * 8);
    const uint32_t ow[4] = {o4.x, o4.y, o4.z, o4.w};
    uint32_t ov[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      ov[2 * j] = __builtin_amdgcn_perm(ow[j], ow[j], 0x01000100u);
      ov[2 * j + 1] = __builtin_amdgcn_perm(ow[j], ow[j], 0x03020302u);
    }
    uint32_t m[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      // narrow to int16 (pack keeps the low halves), clip to [0, smax]
      sp_v2s pk;
      pk.x = (short)va[j];
      pk.y = (short)vb[j];
      pk = __builtin_elementwise_min(__builtin_elementwise_max(pk, sp_s2(0u)), sp_s2(smax2));
      m[j] = sp_u(sp_s2(ov[j]) - pk);
    }
    // vertical WHT (down the column) in registers
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
      for (int i = 0; i < 8; i += len << 1)
#pragma unroll
        for (int j = i; j < i + len; j++) {
          const sp_v2s a = sp_s2(m[j]), b = sp_s2(m[j + len]);
          m[j] = sp_u(a + b);
          m[j + len] = sp_u(a - b);
        }
    // horizontal WHT across the 8 lanes of the tile: stages xor 1, xor 2
    {
      const uint32_t sg = (col & 1) ? 0xffffffffu : 0x00010001u;
#pragma unroll
      for (int j = 0; j < 8; j++) m[j] = sp_pk_mad(m[j], sg, sp_swizzle<0x041F>(m[j]));
    }
    {
      const uint32_t sg = (col & 2) ? 0xffffffffu : 0x00010001u;
#pragma unroll
      for (int j = 0; j < 8; j++) m[j] = sp_pk_mad(m[j], sg, sp_swizzle<0x081F>(m[j]));
    }
    // last stage (xor 4) folded into the absolute sum: 2 * max(|a|, |b|),
    // counted once by each lane of the pair
    uint32_t mx[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const sp_v2s v = sp_s2(m[j]);
      const sp_v2u av = __builtin_bit_cast(sp_v2u, __builtin_elementwise_max(v, sp_s2(0u) - v));
      const sp_v2u ot = __builtin_bit_cast(
          sp_v2u, sp_swizzle<0x101F>(__builtin_bit_cast(uint32_t, av)));
      mx[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(av, ot));
    }
    uint32_t sa = 0, sb = 0;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const sp_v2u t2 = __builtin_bit_cast(sp_v2u, mx[j]) + __builtin_bit_cast(sp_v2u, mx[j + 1]);
      sa = __builtin_amdgcn_udot2(t2, (sp_v2u){1, 0}, sa, false);
      sb = __builtin_amdgcn_udot2(t2, (sp_v2u){0, 1}, sb, false);
    }
    // tile totals over the 8 columns
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0xB1, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0xB1, 0xF, 0xF, false);
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0x4E, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0x4E, 0xF, 0xF, false);
    sa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, 0x141, 0xF, 0xF, false);
    sb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sb, 0x141, 0xF, 0xF, false);
    if (active && col == 0) {
      atomicAdd(&dist[ca], (sa + 2) >> 2);
      if (cb != ca) atomicAdd(&dist[cb], (sb + 2) >> 2);
    }
  }
}

#endif  // XVCGPU_K_SUBPEL_H_
