// k_subpel.h -- T3 fast path: the sub-pel SATD sweep for blocks with both
// sides >= 8 (8x8, 16x8 or 8x16 SATD tiles) at bit depth <= 10, on packed
// 16-bit math.
// Same arithmetic as me2_build_hplanes / me2_satd_cands in k_me2.h
// (inter_prediction.cc:1207-1448, sample_metric.cc:316-641), fewer VALU
// instructions:
//   * the filtered planes are kept COLUMN-major in LDS, so the vertical 8-tap
//     filter of 8 outputs of one column reads 16 consecutive int16 (two
//     ds_read_b128) and is <= 5 v_dot2_i32_i16 per output: rows are consumed as
//     aligned pairs, an odd start uses the tap set shifted by one
//     ((0,t0),(t1,t2),(t3,t4),(t5,t6),(t7,0));
//   * taps (and the unfiltered plane) are pre-scaled by powers of two so that
//     the rounded, shifted prediction is the accumulator's UPPER half: no shift
//     per output, and two candidates' outputs are packed by one v_perm_b32;
//   * lane = one column of one 8x8 tile for TWO candidates, one in each
//     16-bit half: clip, residual and the Hadamard butterflies are v_pk_*;
//     the vertical WHT is in registers, the horizontal one goes through
//     ds_swizzle (LDS crossbar, no VALU);
//   * the candidates of a pass are paired by what they share (kSubpelOrder, k_me2.h):
//     the two half-pel candidates above / below a position are ONE filtered
//     column read at two row offsets (9 outputs instead of 16), a candidate
//     whose vertical phase is zero is one tap, not eight;
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
// the same with the addend in a register of its own: the compiler's two-address
// form (v_dot2c) would copy a live addend (the rounding offset) first
__device__ __forceinline__ int sp_dot2_from(uint32_t a, uint32_t b, int c) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
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
// (a.lo - b.lo, a.lo - b.hi) / (a.hi - b.lo, a.hi - b.hi): one original sample
// against the two candidates' predictions, the replication done by op_sel
__device__ __forceinline__ uint32_t sp_pk_sub_lo(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t sp_pk_sub_hi(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_sub_i16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// upper halves of two accumulators side by side: (a >> 16) | (b & 0xffff0000)
__device__ __forceinline__ uint32_t sp_pack_hi(int a, int b) {
  return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x07060302u);
}
template <int PATTERN>
__device__ __forceinline__ uint32_t sp_swizzle(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, PATTERN);
}

// Per-candidate parameters of the vertical stage, 16 dwords.  The taps are scaled by
// 2^k, k chosen per kind of candidate so that (sum + off) >> 16 is the prediction:
//   unfiltered plane (x phase 0; the plane holds sample << 4): k = 6, off = 32 << 10
//     (FilterVerSampleSample: (sum + 32) >> 6, identity taps for y phase 0);
//   horizontal only (one tap 2^(bd - 8) on the 14-bit plane, me2_honly_off): k = 10;
//   two-stage (FilterVerShortSample, shift 20 - bd): k = bd - 4.
struct __attribute__((aligned(16))) SpCand {
  uint32_t e[5];   // tap pairs for even outputs of the column
  uint32_t o[5];   // tap pairs for odd outputs
  int off;         // rounding offset, scaled like the taps
  int plane;       // int16 index of the plane's (this candidate's x = 0, stored row 0)
  int q;           // stored row of the first tap for output row 0: 0 or 1
  int ident;       // vertical phase 0 (or horizontal only): one tap, at index 3
  int pad[2];
};

// The sweep takes the candidates two at a time, (2 p, 2 p + 1) of the order their records
// are stored in (me2_subpel_fast_pass stores them so that candidates which share work are
// neighbours), and record 2 p says what the pair shares:
//   SP_ROWSHARED: same column, same taps, rows one apart (the half-pel pass's
//     above / below: one filtered column of nine outputs read at two offsets);
//   SP_IDENT: both candidates' vertical phase is zero (one tap each);
//   SP_GENERIC: two independent vertical filters.
// A record is 12 dwords: e[0..4], o[0..4], off, plane | kind << 24.
enum { SP_GENERIC = 0, SP_ROWSHARED = 1, SP_IDENT = 2 };
#define SP_REC 12
__device__ __forceinline__ void sp_store_cand(uint32_t *rec, const SpCand &c, int kind) {
  uint4 *r = reinterpret_cast<uint4 *>(rec);
  r[0] = make_uint4(c.e[0], c.e[1], c.e[2], c.e[3]);
  r[1] = make_uint4(c.e[4], c.o[0], c.o[1], c.o[2]);
  r[2] = make_uint4(c.o[3], c.o[4], (uint32_t)c.off, (uint32_t)c.plane | ((uint32_t)kind << 24));
}

// Column-major plane of one x-phase: p14 = FilterHorSampleShort (14 bit), or
// the unfiltered samples << 4 when fx == 0; `ncols` (even) columns from picture
// column pel_x on; column stride h + 8, stored row r <-> picture row r - 4
// relative to the full-pel position.  win: row-major window, cols -8..w+7
// (row stride w + 16).
__device__ __forceinline__ void sp_build_planes(const uint16_t *win, int16_t *p14,
                                                const int16_t (*taps)[8],
                                                int bd, int w, int h, int pel_x, int fx,
                                                int ncols, int tid = threadIdx.x & 63,
                                                int nthr = 64) {
  // tid / nthr: this thread's place in the wave (or team of waves).  A thread keeps its pair
  // of columns and walks down the rows (the threads cover nthr / (w / 2) rows at a time): the
  // window and plane addresses advance by constants.  The two extra columns of a wide plane
  // (ncols = w + 2) are a second walk, one row per thread.
  const int ws = w + 16, rs = h + 8;
  const int hw = w >> 1, lhw = 31 - __clz(hw);
  const int rstep = nthr >> lhw;                       // w <= 64: hw <= 32 <= nthr
  const int shift = 6 - (14 - bd), offset = -(8192 << shift);
  const bool odd = ((pel_x + 5) & 1) != 0;             // parity of the first tap's window column
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
  if (fx != 0) {
    const int16_t *f = taps[fx];  // LDS copy of kLumaTaps
    a0 = sp_pack_taps(f[0], f[1]); a1 = sp_pack_taps(f[2], f[3]);
    a2 = sp_pack_taps(f[4], f[5]); a3 = sp_pack_taps(f[6], f[7]);
    b0 = sp_pack_taps(0, f[0]); b1 = sp_pack_taps(f[1], f[2]); b2 = sp_pack_taps(f[3], f[4]);
    b3 = sp_pack_taps(f[5], f[6]); b4 = sp_pack_taps(f[7], 0);
  }
  // one (row, pair of columns x0, x0 + 1): src = the window row
  auto unit = [&](const uint16_t *wrow, int x0, int16_t *dst) {
    if (fx == 0) {
      const uint16_t *src = wrow + x0 + pel_x + 8;
      dst[0] = (int16_t)(src[0] << 4);
      dst[rs] = (int16_t)(src[1] << 4);
      return;
    }
    // first tap of output x0 sits at window column c0 = x0 + pel_x + 5
    const uint32_t *d = reinterpret_cast<const uint32_t *>(wrow) + ((x0 + pel_x + 5) >> 1);
    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
    int s0, s1;
    if (!odd) {
      s0 = sp_dot2_from(d0, a0, offset); s0 = sp_dot2(d1, a1, s0); s0 = sp_dot2(d2, a2, s0);
      s0 = sp_dot2(d3, a3, s0);
      s1 = sp_dot2_from(d0, b0, offset); s1 = sp_dot2(d1, b1, s1); s1 = sp_dot2(d2, b2, s1);
      s1 = sp_dot2(d3, b3, s1); s1 = sp_dot2(d4, b4, s1);
    } else {
      s0 = sp_dot2_from(d0, b0, offset); s0 = sp_dot2(d1, b1, s0); s0 = sp_dot2(d2, b2, s0);
      s0 = sp_dot2(d3, b3, s0); s0 = sp_dot2(d4, b4, s0);
      s1 = sp_dot2_from(d1, a0, offset); s1 = sp_dot2(d2, a1, s1); s1 = sp_dot2(d3, a2, s1);
      s1 = sp_dot2(d4, a3, s1);
    }
    dst[0] = (int16_t)(s0 >> shift);
    dst[rs] = (int16_t)(s1 >> shift);
  };
  {
    const int x0 = (tid & (hw - 1)) << 1;
    int r = tid >> lhw;
    const uint16_t *wrow = win + r * ws;
    int16_t *dst = p14 + x0 * rs + r;
    if (fx != 0) {
      // three rows per trip, their window reads issued together (a 16x16 block's plane is
      // one trip of a wave)
      const int cw = (x0 + pel_x + 5) >> 1;
      for (; r + 2 * rstep < rs; r += 3 * rstep, wrow += 3 * rstep * ws, dst += 3 * rstep) {
        uint32_t d[3][5];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const uint32_t *dp = reinterpret_cast<const uint32_t *>(wrow + k * rstep * ws) + cw;
#pragma unroll
          for (int j = 0; j < 5; j++) d[k][j] = dp[j];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          int s0, s1;
          if (!odd) {
            s0 = sp_dot2_from(d[k][0], a0, offset); s0 = sp_dot2(d[k][1], a1, s0);
            s0 = sp_dot2(d[k][2], a2, s0); s0 = sp_dot2(d[k][3], a3, s0);
            s1 = sp_dot2_from(d[k][0], b0, offset); s1 = sp_dot2(d[k][1], b1, s1);
            s1 = sp_dot2(d[k][2], b2, s1); s1 = sp_dot2(d[k][3], b3, s1);
            s1 = sp_dot2(d[k][4], b4, s1);
          } else {
            s0 = sp_dot2_from(d[k][0], b0, offset); s0 = sp_dot2(d[k][1], b1, s0);
            s0 = sp_dot2(d[k][2], b2, s0); s0 = sp_dot2(d[k][3], b3, s0);
            s0 = sp_dot2(d[k][4], b4, s0);
            s1 = sp_dot2_from(d[k][1], a0, offset); s1 = sp_dot2(d[k][2], a1, s1);
            s1 = sp_dot2(d[k][3], a2, s1); s1 = sp_dot2(d[k][4], a3, s1);
          }
          dst[k * rstep] = (int16_t)(s0 >> shift);
          dst[k * rstep + rs] = (int16_t)(s1 >> shift);
        }
      }
    }
    for (; r < rs; r += rstep, wrow += rstep * ws, dst += rstep) unit(wrow, x0, dst);
  }
  if (ncols > w)
    for (int r = tid; r < rs; r += nthr) unit(win + r * ws, w, p14 + w * rs + r);
}

// Tap sets of a candidate: q = (stored row of its first tap for output row 0)
// in {0, 1}; fy = row of `taps` (vertical phase; 0 -> identity taps; 16 -> the
// horizontal-only tap); ks = log2 of the scale.
__device__ __forceinline__ void sp_fill_taps(SpCand &c, const int16_t (*taps)[8], int fy,
                                             int q, int ks) {
  const int16_t *t = taps[fy];
  int s[8];
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = (int)t[i] << ks;
  const uint32_t a0 = sp_pack_taps(s[0], s[1]), a1 = sp_pack_taps(s[2], s[3]),
                 a2 = sp_pack_taps(s[4], s[5]), a3 = sp_pack_taps(s[6], s[7]);
  const uint32_t b0 = sp_pack_taps(0, s[0]), b1 = sp_pack_taps(s[1], s[2]),
                 b2 = sp_pack_taps(s[3], s[4]), b3 = sp_pack_taps(s[5], s[6]),
                 b4 = sp_pack_taps(s[7], 0);
  if (q == 0) {
    c.e[0] = a0; c.e[1] = a1; c.e[2] = a2; c.e[3] = a3; c.e[4] = 0;
    c.o[0] = b0; c.o[1] = b1; c.o[2] = b2; c.o[3] = b3; c.o[4] = b4;
  } else {
    c.e[0] = b0; c.e[1] = b1; c.e[2] = b2; c.e[3] = b3; c.e[4] = b4;
    c.o[0] = 0; c.o[1] = a0; c.o[2] = a1; c.o[3] = a2; c.o[4] = a3;
  }
  c.q = q;
}

// SATD of `ncand` candidates (records in cand[]) against origc (the
// original block, column-major, column stride h) with TW x TH tiles: 8x8 (square
// blocks), 16x8 (w > h) or 8x16 (w < h), ComputeSatdNxM's choice for blocks
// with both sides >= 8 (sample_metric.cc:403-641).  Adds the normalised tile
// sums into dist[c].  w, h powers of two, multiples of TW, TH; bd <= 10.
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

// 16 stored rows (8 aligned pairs) of one plane column
__device__ __forceinline__ void sp_load_col(const int16_t *lds, int idx, uint32_t p[8]) {
  const uint4 *src = reinterpret_cast<const uint4 *>(lds + idx);
  const uint4 lo = src[0], hi = src[1];
  p[0] = lo.x; p[1] = lo.y; p[2] = lo.z; p[3] = lo.w;
  p[4] = hi.x; p[5] = hi.y; p[6] = hi.z; p[7] = hi.w;
}

// the three ways to the packed predictions pk[8] (low half: candidate a) of a tile column;
// pa / pb: the 16 stored rows of the candidates' plane columns, ta / tb: their e[5], o[5]
__device__ __forceinline__ void sp_pred_generic(const uint32_t pa[8], const uint32_t pb[8],
                                                const uint32_t *ta, const uint32_t *tb,
                                                int off_a, int off_b, uint32_t pk[8]) {
#pragma unroll
  for (int m = 0; m < 4; m++) {
    int ea = sp_dot2_from(pa[m], ta[0], off_a), oa = sp_dot2_from(pa[m], ta[5], off_a);
    int eb = sp_dot2_from(pb[m], tb[0], off_b), ob = sp_dot2_from(pb[m], tb[5], off_b);
#pragma unroll
    for (int i = 1; i < 5; i++) {
      ea = sp_dot2(pa[m + i], ta[i], ea);
      oa = sp_dot2(pa[m + i], ta[5 + i], oa);
      eb = sp_dot2(pb[m + i], tb[i], eb);
      ob = sp_dot2(pb[m + i], tb[5 + i], ob);
    }
    pk[2 * m] = sp_pack_hi(ea, eb);
    pk[2 * m + 1] = sp_pack_hi(oa, ob);
  }
}
// candidate b = candidate a one row further down the same filtered column
// (q_a = 0: e = the aligned tap pairs, e[4] = 0; o = the shifted ones)
__device__ __forceinline__ void sp_pred_rowshared(const uint32_t pa[8], const uint32_t *ta,
                                                  int off_a, uint32_t pk[8]) {
  int out[9];
#pragma unroll
  for (int m = 0; m < 5; m++) {
    int e = sp_dot2_from(pa[m], ta[0], off_a);
#pragma unroll
    for (int i = 1; i < 4; i++) e = sp_dot2(pa[m + i], ta[i], e);
    out[2 * m] = e;
    if (m < 4) {
      int o = sp_dot2_from(pa[m], ta[5], off_a);
#pragma unroll
      for (int i = 1; i < 5; i++) o = sp_dot2(pa[m + i], ta[5 + i], o);
      out[2 * m + 1] = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) pk[j] = sp_pack_hi(out[j], out[j + 1]);
}
// one tap each, at index 3: output j reads stored row j + q + 3 - pair m + 1 (q = 0) or
// m + 2 (q = 1) for the even outputs (the other set's tap pair is zero), pair m + 2 for the
// odd ones
__device__ __forceinline__ void sp_pred_ident(const uint32_t pa[8], const uint32_t pb[8],
                                              const uint32_t *ta, const uint32_t *tb, int off_a,
                                              int off_b, uint32_t pk[8]) {
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int ea = sp_dot2(pa[m + 2], ta[2], sp_dot2_from(pa[m + 1], ta[1], off_a));
    const int eb = sp_dot2(pb[m + 2], tb[2], sp_dot2_from(pb[m + 1], tb[1], off_b));
    const int oa = sp_dot2_from(pa[m + 2], ta[7], off_a);
    const int ob = sp_dot2_from(pb[m + 2], tb[7], off_b);
    pk[2 * m] = sp_pack_hi(ea, eb);
    pk[2 * m + 1] = sp_pack_hi(oa, ob);
  }
}

template <int TW, int TH>
__device__ __forceinline__ void sp_satd_pairs_t(const int16_t *lds, const uint32_t *cand,
                                                int ncand, const uint16_t *origc,
                                                uint32_t *dist, int bd, int w, int h, int tid,
                                                int nthr) {
  static_assert((TW == 8 && TH == 8) || (TW == 16 && TH == 8) || (TW == 8 && TH == 16), "tile");
  constexpr int LT = TW == 16 ? 4 : 3;          // stages across lanes
  constexpr int HF = TH == 16 ? 1 : 2;          // of which formed as values
  constexpr bool MAG = LT - HF == 2;            // one magnitude-only stage before the fold
  const int lane = threadIdx.x & 63;
  const int rs = h + 8;
  const int tiles_x = w / TW, ltx = 31 - __clz(tiles_x);
  const int upp = (w * h) / TH, lupp = 31 - __clz(upp);   // lanes (tile columns) per pair
  const int total = upp * ((ncand + 1) >> 1);
  const uint32_t smax2 = (uint32_t)((1 << bd) - 1) * 0x10001u;
  for (int g0 = tid & ~63; g0 < total; g0 += nthr) {   // whole waves stay in step
    const int g = g0 + lane;
    const bool active = g < total;
    const int gg = active ? g : 0;
    const int pr = gg >> lupp, u = gg & (upp - 1);
    const int tile = u / TW, col = u & (TW - 1);
    const int tx = tile & (tiles_x - 1), ty = tile >> ltx;
    const int x = tx * TW + col;
    const int ca = 2 * pr, cb = (2 * pr + 1 < ncand) ? 2 * pr + 1 : 2 * pr;
    // round trip 1: the two records; round trip 2: the two plane columns and the originals
    uint32_t ta[12], tb[12];
    {
      const uint4 *ra = reinterpret_cast<const uint4 *>(cand + ca * SP_REC);
      const uint4 *rb = reinterpret_cast<const uint4 *>(cand + cb * SP_REC);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const uint4 va = ra[i], vb = rb[i];
        ta[4 * i] = va.x; ta[4 * i + 1] = va.y; ta[4 * i + 2] = va.z; ta[4 * i + 3] = va.w;
        tb[4 * i] = vb.x; tb[4 * i + 1] = vb.y; tb[4 * i + 2] = vb.z; tb[4 * i + 3] = vb.w;
      }
    }
    const int kind = (int)(ta[11] >> 24);
    const int plane_a = (int)(ta[11] & 0xffffffu), plane_b = (int)(tb[11] & 0xffffffu);
    const int off_a = (int)ta[10], off_b = (int)tb[10];
    const int col_off = x * rs + ty * TH;
    uint32_t m[TH];
#pragma unroll
    for (int r8 = 0; r8 < TH / 8; r8++) {
      uint32_t pa[8], pb[8], pk[8];
      sp_load_col(lds, plane_a + col_off + 8 * r8, pa);
      sp_load_col(lds, plane_b + col_off + 8 * r8, pb);
      const uint4 o4 = *reinterpret_cast<const uint4 *>(origc + x * h + ty * TH + 8 * r8);
      if (kind == SP_ROWSHARED) sp_pred_rowshared(pa, ta, off_a, pk);
      else if (kind == SP_IDENT) sp_pred_ident(pa, pb, ta, tb, off_a, off_b, pk);
      else sp_pred_generic(pa, pb, ta, tb, off_a, off_b, pk);
      const uint32_t ow[4] = {o4.x, o4.y, o4.z, o4.w};   // 8 originals of the column
#pragma unroll
      for (int j = 0; j < 8; j++) {
        // clip to [0, smax], residual of both candidates
        const sp_v2s c2 = __builtin_elementwise_min(
            __builtin_elementwise_max(sp_s2(pk[j]), sp_s2(0u)), sp_s2(smax2));
        m[8 * r8 + j] = (j & 1) ? sp_pk_sub_hi(ow[j >> 1], sp_u(c2)) : sp_pk_sub_lo(ow[j >> 1], sp_u(c2));
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
    // horizontal WHT across the TW lanes of the tile: formed stages.  The exchanges of a
    // stage are issued together and waited for once (left to itself the compiler, short of
    // registers, pairs every ds_swizzle with its own wait: a crossbar round trip per value).
#define SP_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SP_WAIT_LGKM() __builtin_amdgcn_s_waitcnt(0xc07f)
    {
      const uint32_t sg = (col & 1) ? 0xffffffffu : 0x00010001u;
      uint32_t t[TH];
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) t[j] = sp_swz_xor<1>(m[j]);
      SP_WAIT_LGKM();
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) m[j] = sp_pk_mad(m[j], sg, t[j]);
    }
    if (HF == 2) {
      const uint32_t sg = (col & 2) ? 0xffffffffu : 0x00010001u;
      uint32_t t[TH];
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) t[j] = sp_swz_xor<2>(m[j]);
      SP_WAIT_LGKM();
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) m[j] = sp_pk_mad(m[j], sg, t[j]);
    }
    uint32_t sa = 0, sb = 0;
    if (!MAG) {
      // last stage (xor 4) folded into the absolute sum: 2 * max(|a|, |b|),
      // counted once by each lane of the pair
      uint32_t av[TH], ot[TH];
#pragma unroll
      for (int j = 0; j < TH; j++) av[j] = sp_pk_abs(m[j]);
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) ot[j] = sp_swz_xor<4>(av[j]);
      SP_WAIT_LGKM();
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j += 2) {
        const sp_v2u t2 =
            __builtin_elementwise_max(__builtin_bit_cast(sp_v2u, av[j]), __builtin_bit_cast(sp_v2u, ot[j])) +
            __builtin_elementwise_max(__builtin_bit_cast(sp_v2u, av[j + 1]), __builtin_bit_cast(sp_v2u, ot[j + 1]));
        sa = __builtin_amdgcn_udot2(t2, (sp_v2u){1, 0}, sa, false);
        sb = __builtin_amdgcn_udot2(t2, (sp_v2u){0, 1}, sb, false);
      }
    } else {
      constexpr int KM = 1 << HF, KF = 2 << HF;   // magnitude stage, fold stage
      // this lane forms u + v (bit clear) or v - u (bit set) of the pair
      const uint32_t flip = (col & KM) ? 0xffffffffu : 0u;
      uint32_t vv[TH], mag[TH], ot[TH];
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) vv[j] = sp_swz_xor<KM>(m[j]);
      SP_WAIT_LGKM();
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) {
        const uint32_t uu = m[j];
        const sp_v2u au = __builtin_bit_cast(sp_v2u, sp_pk_abs(uu));
        const sp_v2u av = __builtin_bit_cast(sp_v2u, sp_pk_abs(vv[j]));
        const uint32_t sum = __builtin_bit_cast(uint32_t, au + av);
        const uint32_t dif = __builtin_bit_cast(
            uint32_t, __builtin_elementwise_max(au, av) - __builtin_elementwise_min(au, av));
        // halves whose signs differ (all-ones), swapped for the subtracting lane
        const sp_v2s sx = sp_s2(uu ^ vv[j]) >> (sp_v2s){15, 15};
        const uint32_t sel = sp_u(sx) ^ flip;
        mag[j] = (dif & sel) | (sum & ~sel);
      }
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) ot[j] = sp_swz_xor<KF>(mag[j]);
      SP_WAIT_LGKM();
      SP_STAGE_FENCE();
#pragma unroll
      for (int j = 0; j < TH; j++) {
        const sp_v2u mx = __builtin_elementwise_max(__builtin_bit_cast(sp_v2u, mag[j]),
                                                    __builtin_bit_cast(sp_v2u, ot[j]));
        sa = __builtin_amdgcn_udot2(mx, (sp_v2u){1, 0}, sa, false);
        sb = __builtin_amdgcn_udot2(mx, (sp_v2u){0, 1}, sb, false);
      }
    }
#undef SP_STAGE_FENCE
#undef SP_WAIT_LGKM
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

__device__ __forceinline__ void sp_satd_pairs(const int16_t *lds, const uint32_t *cand,
                                              int ncand, const uint16_t *origc, uint32_t *dist,
                                              int bd, int w, int h,
                                              int tid = threadIdx.x & 63, int nthr = 64) {
  if (w == h) sp_satd_pairs_t<8, 8>(lds, cand, ncand, origc, dist, bd, w, h, tid, nthr);
  else if (w > h) sp_satd_pairs_t<16, 8>(lds, cand, ncand, origc, dist, bd, w, h, tid, nthr);
  else sp_satd_pairs_t<8, 16>(lds, cand, ncand, origc, dist, bd, w, h, tid, nthr);
}

#endif  // XVCGPU_K_SUBPEL_H_
