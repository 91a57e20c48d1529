// k_me2.h -- T1 + T3 (+W1, I1, M1, M4), throughput form: ONE WAVEFRONT per
// motion-estimation job (= one InterSearch::MotionEstNormal call,
// inter_search.cc:606-662).  A 1080p picture has 8160 such jobs: with one wave
// each the whole picture is resident at once (32 waves/CU x 256 CUs) and the
// serial phase structure of the search is hidden by occupancy instead of being
// paid in barriers.  No workgroup barrier is needed (a 64-thread workgroup is
// one wave; __syncthreads() only orders LDS traffic).
//
// Full-pel (TzSearch::Search, inter_tz_search.cc:84-171): the candidate
// pattern is a table (tz_pattern.h), lane i owns entries i and i+64; quads of
// 4 lanes evaluate 16 candidates per pass (16-byte unaligned loads, v_sad_u16,
// quad_perm DPP sums); costs go through LDS so that lane i gets the costs of
// its own candidates, and the reference's ordered folds become wave-wide keyed
// minima ((cost << 7) | index: lowest index wins ties, exactly the strict-<
// left-to-right fold).  The initial predictor / zero / previous-MV checks are
// one pass; the rare step-5 grid runs one candidate per lane.
//
// Sub-pel (InterSearch::SubpelSearch, inter_search.cc:893-964): the reference
// window (block + 8-tap support + 1 pel) is staged once in LDS; for every
// distinct horizontal phase of the 9 (then 8) candidates the horizontally
// filtered planes are built once (14-bit intermediate for the two-stage path,
// Sample-rounded for the horizontal-only path); every candidate is then a
// vertical 8-tap over one of those planes - identity taps reproduce the copy /
// horizontal-only paths bit-exactly - evaluated directly in SATD tile layout
// (one lane = one tile row in registers, vertical WHT by DPP/swizzle), so no
// predicted block is ever stored.
#ifndef XVCGPU_K_ME2_H_
#define XVCGPU_K_ME2_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_me.h"
#include "k_subpel.h"
#include "xvcgpu_internal.h"

// Developer build (-DXVCGPU_TRACE): per-job phase timestamps (s_memtime) of
// the ME kernel, one row per job, plain stores (tools/trace_me.py).
#ifdef XVCGPU_TRACE
__device__ unsigned long long g_me2_trace[32768][24];
#define ME2_TRACE(k)                                                        \
  do {                                                                      \
    if ((threadIdx.x & 63) == 0 && bi < 32768)                              \
      g_me2_trace[bi][k] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
// wall clock (100 MHz, common to the whole device) for cross-wave timelines
#define ME2_TRACE_RT(k)                                                     \
  do {                                                                      \
    if ((threadIdx.x & 63) == 0 && bi < 32768)                              \
      g_me2_trace[bi][k] = __builtin_amdgcn_s_memrealtime();                \
  } while (0)
// event counters of a job (columns 11..15): diamond sweeps, 16-candidate passes,
// refinement iterations, neighbour steps, candidates evaluated
// columns 16..23: clocks spent in the sections of the sub-pel passes (ME2_CLK)
#define ME2_COUNT_DECL int me2_cnt[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ME2_COUNT(k, v) me2_cnt[k] += (v)
#define ME2_COUNT_STORE()                                                   \
  do {                                                                      \
    if ((threadIdx.x & 63) == 0 && bi < 32768)                              \
      for (int q_ = 0; q_ < 13; q_++) g_me2_trace[bi][11 + q_] = (unsigned long long)me2_cnt[q_]; \
  } while (0)
#define ME2_CLK_ARG , int *me2_cnt
#define ME2_CLK_PASS , me2_cnt
#define ME2_CLK_BEGIN unsigned long long me2_t0 = __builtin_amdgcn_s_memtime()
#define ME2_CLK(k)                                                          \
  do {                                                                      \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();             \
    me2_cnt[5 + (k)] += (int)(t_ - me2_t0);                                 \
    me2_t0 = t_;                                                            \
  } while (0)
#else
#define ME2_TRACE(k) do {} while (0)
#define ME2_TRACE_RT(k) do {} while (0)
#define ME2_COUNT_DECL do {} while (0)
#define ME2_COUNT(k, v) do {} while (0)
#define ME2_COUNT_STORE() do {} while (0)
#define ME2_CLK_ARG
#define ME2_CLK_PASS
#define ME2_CLK_BEGIN do {} while (0)
#define ME2_CLK(k) do {} while (0)
#endif

// ---- cross-lane helpers (DPP / swizzle; no LDS traffic) ---------------------
template <int S>
__device__ __forceinline__ int lane_xor(int v) {
  if (S == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
  if (S == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
  if (S == 4) return __builtin_amdgcn_ds_swizzle(v, 0x101F);
  if (S == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);
  if (S == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);
  return __shfl_xor(v, 32, 64);
}

// Sum over each aligned group of G lanes (G in {4, 8, 16}); all lanes get it.
template <int G>
__device__ __forceinline__ int dpp_group_sum(int v) {
  v += lane_xor<1>(v);
  v += lane_xor<2>(v);
  if (G >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
  if (G >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
  return v;
}

// Wave-wide minimum of a 32-bit key, returned wave-uniform.
__device__ __forceinline__ uint32_t wave_min_key(uint32_t v) {
  uint32_t o;
  o = (uint32_t)lane_xor<1>((int)v); v = v < o ? v : o;
  o = (uint32_t)lane_xor<2>((int)v); v = v < o ? v : o;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
  v = v < o ? v : o;
  o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
  v = v < o ? v : o;
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
  const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
  const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}

// Wave-wide integer sum, returned wave-uniform (all 64 lanes participate).
__device__ __forceinline__ int wave_reduce_add_i32(int v) {
  v = dpp_group_sum<16>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
         __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

#define ME2_NOKEY 0xffffffffu
// independent jobs (waves) per workgroup, by block-size class (LDS budget)
#ifndef ME2_WAVES16
#define ME2_WAVES16 1
#endif
#define ME2_WAVES(MS) ((MS) > 32 ? 2 : ((MS) > 16 ? 4 : ME2_WAVES16))
// waves per SIMD the register allocator leaves room for: the 16-class fits 4
// workgroups per CU by LDS, so cap its registers at 128 (measured +3 %)
#ifndef ME2_SQ16_MIN_WAVES
#define ME2_SQ16_MIN_WAVES 5
#endif
#define ME2_MIN_WAVES(MS) ((MS) <= 16 ? 4 : 1)  // 5 (96 VGPRs) spills: 80 -> 147 us

// Orders LDS traffic between the lanes of ONE wave (jobs never share data
// across waves, so no workgroup barrier is ever needed): LDS requests of a
// wave are serviced in issue order; this only stops compiler reordering and
// drains outstanding LDS returns.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#define ME2_LANE ((int)(threadIdx.x & 63))

// H-only sub-pel prediction from the 14-bit plane.  FilterHorSampleSample
// (inter_prediction.cc) is clip((S + 32) >> 6) of the 8-tap sum S; the plane
// holds v = (S - (8192 << t)) >> t with t = bd - 8, so (v << t) + (8192 << t)
// = S - (S mod 2^t), and since 2^t divides 32 the dropped bits never carry
// into bit 6: ((v << t) + (8192 << t) + 32) >> 6 == (S + 32) >> 6 exactly.
// That is the candidate filter with a single centre tap 2^t (row 16 of
// Me2SharedT::taps), offset 32 + (8192 << t), shift 6 - no second plane.
__device__ __forceinline__ int me2_honly_off(int bd, int cfx) {
  return cfx != 0 ? 32 + (8192 << (bd - 8)) : 32;
}
__device__ __forceinline__ int me2_vtaps_row(int cfx, int cfy) {
  return (cfx != 0 && cfy == 0) ? 16 : cfy;
}

// SUB = false: layout of a full-pel-only kernel instance (orig + cost).
template <int MS, bool SUB>
struct __attribute__((aligned(16))) Me2SharedT {
  uint16_t orig[MS * MS];                   // row stride w
  uint16_t win[SUB ? (MS + 8) * (MS + 16) : 8];  // rows -4..h+3, cols -8..w+7
  // per x-phase slot: 14-bit H-filtered plane, rows -4..h+3 (the fast path keeps
  // the unfiltered samples here for a phase-0 slot).  The Sample-rounded
  // H-only prediction is derived from the 14-bit value, see me2_honly_taps.
  // (a slot is two columns wider than the block: the half-pel pass's x phase 8 is needed from
  // one column to the left AND from the block's own column - one plane of w + 2 columns)
  // slot k holds x phase 4 k (the phases of a search are multiples of a quarter pel)
  int16_t hint[4][SUB ? (MS + 8) * (MS + 2) : 8];
  uint32_t cost[128];
  // per sub-pel candidate: plane offset (int16 units from `orig`), stride,
  // rounding offset, shift, 8 taps
  union {
    struct {                                // row-major path
      int cand_plane[12], cand_stride[12], cand_off[12], cand_shift[12];
      int16_t cand_taps[12][8];
    };
    uint32_t rec[SUB ? 10 * SP_REC : 4];    // fast path (k_subpel.h): a record per candidate
  };
  uint32_t dist[12];
  int dsum[12];                             // AC-only: sum(orig - pred) per candidate
  int16_t taps[SUB ? 17 : 1][8];            // LDS copy of kLumaTaps + the H-only row
};
template <int MS>
using Me2Shared = Me2SharedT<MS, true>;

// ---- full-pel SAD evaluation: 4 lanes per candidate ------------------------
// s.cost[0..n) holds packed positions ((y << 16) | (x & 0xffff), or
// ME2_NOPOS); on return s.cost[i] is the kSad / kSadFast value of candidate i.
// The block's 8-sample (4 for w == 4) segments are dealt round-robin to the 4
// lanes of a quad (segment sg -> lane sg & 3), which keeps the two halves of a
// 16-wide row - or 4 rows of an 8-wide block - in one wave instruction: every
// cache line of a candidate is looked up once, exactly as with one candidate
// per 16 lanes, but 16 candidates are in flight per pass, the reduction is two
// quad_perm DPP adds and a pass costs ~2 instructions per candidate.
#define ME2_NOPOS 0x80008000u

// Candidate rows are addressed as (uniform base) + (32-bit unsigned byte offset): one
// scalar base per job, one multiply-add per candidate and one add per load instead of
// 64-bit address arithmetic per load.  The base is the CU's position in the reference
// plane moved down by ME2_ADDR_BIAS bytes, so that offsets of positions left of / above
// the CU stay positive (a plane is far smaller than the bias).
#define ME2_ADDR_BIAS (1u << 28)
__device__ __forceinline__ const char *me2_ref_base(const MeCtx &c) {
  return reinterpret_cast<const char *>(c.ref) - ME2_ADDR_BIAS;
}
__device__ __forceinline__ uint32_t me2_pos_off(const MeCtx &c, int x, int y) {
  return (uint32_t)((y * c.rs + x) * 2) + ME2_ADDR_BIAS;
}

// This lane's share of the original block for the quad evaluation (SPL segments of 8
// samples) and the byte offsets of those segments in a candidate block.
template <int SPL>
struct Me2Quad {
  uint4 a[SPL];
  uint32_t goff[SPL];
};
template <int SPL>
__device__ __forceinline__ void me2_quad_load(const MeCtx &c, const uint16_t *s_orig,
                                              Me2Quad<SPL> &qd) {
  const int q = ME2_LANE & 3;
  const int spr = c.w >> 3, lspr = 31 - __clz(spr);
#pragma unroll
  for (int u = 0; u < SPL; u++) {
    const int sg = q + 4 * u;
    const int y = (sg >> lspr) * c.row_step, x = (sg & (spr - 1)) << 3;
    qd.a[u] = *reinterpret_cast<const uint4 *>(s_orig + y * c.w + x);
    qd.goff[u] = (uint32_t)((y * c.rs + x) * 2);
  }
}

// SAD of the candidate at packed position pk against this lane's segments, summed over the
// quad (valid in all four lanes).  pk == ME2_NOPOS: nothing is read, 0.
template <int SPL>
__device__ __forceinline__ uint32_t me2_quad_sad(const MeCtx &c, const Me2Quad<SPL> &qd,
                                                 const char *base, uint32_t pk) {
  uint32_t sum = 0;
  if (pk != ME2_NOPOS) {  // uniform within the quad
    const int x = (int)(int16_t)(pk & 0xffffu), y = (int)pk >> 16;
    const uint32_t vo = me2_pos_off(c, x, y);
    U16x8 b[SPL];
#pragma unroll
    for (int u = 0; u < SPL; u++)
      b[u] = *reinterpret_cast<const U16x8 *>(base + (size_t)(vo + qd.goff[u]));
#pragma unroll
    for (int u = 0; u < SPL; u++) {
      sum = __builtin_amdgcn_sad_u16(qd.a[u].x, b[u].v[0], sum);
      sum = __builtin_amdgcn_sad_u16(qd.a[u].y, b[u].v[1], sum);
      sum = __builtin_amdgcn_sad_u16(qd.a[u].z, b[u].v[2], sum);
      sum = __builtin_amdgcn_sad_u16(qd.a[u].w, b[u].v[3], sum);
    }
  }
  sum += (uint32_t)lane_xor<1>((int)sum);
  sum += (uint32_t)lane_xor<2>((int)sum);
  return (sum * c.sad_mul) >> c.sad_shift;
}

template <int SPL>  // segments per lane, original segments kept in registers
__device__ __forceinline__ void me2_eval_quads_reg(const MeCtx &c, uint32_t *cost,
                                                   const Me2Quad<SPL> &qd, int n) {
  const int lane = ME2_LANE, qi = lane >> 2, q = lane & 3;
  const char *base = me2_ref_base(c);
  for (int i0 = 0; i0 < n; i0 += 16) {
    const int i = i0 + qi;
    const uint32_t pk = i < n ? cost[i] : ME2_NOPOS;
    if (pk == ME2_NOPOS) continue;  // uniform within the quad
    const uint32_t sad = me2_quad_sad<SPL>(c, qd, base, pk);
    if (q == 0) cost[i] = sad;
  }
}

// generic form (large blocks, w == 4): segments streamed from LDS
__device__ __forceinline__ void me2_eval_quads_gen(const MeCtx &c, uint32_t *cost,
                                                   const uint16_t *s_orig, int n) {
  const int lane = ME2_LANE, qd = lane >> 2, q = lane & 3;
  const bool wide = c.w >= 8;
  const int spr = wide ? c.w >> 3 : 1, lspr = 31 - __clz(spr);
  const int nseg = c.rows * spr;
  const char *base = me2_ref_base(c);
  for (int i0 = 0; i0 < n; i0 += 16) {
    const int i = i0 + qd;
    const uint32_t pk = i < n ? cost[i] : ME2_NOPOS;
    if (pk == ME2_NOPOS) continue;
    const int x = (int)(int16_t)(pk & 0xffffu), y = (int)pk >> 16;
    const uint32_t vo = me2_pos_off(c, x, y);
    uint32_t sum = 0;
    if (wide) {
      for (int sg = q; sg < nseg; sg += 4) {
        const int yy = (sg >> lspr) * c.row_step, xx = (sg & (spr - 1)) << 3;
        const uint4 a = *reinterpret_cast<const uint4 *>(s_orig + yy * c.w + xx);
        const U16x8 b = *reinterpret_cast<const U16x8 *>(
            base + (size_t)(vo + (uint32_t)((yy * c.rs + xx) * 2)));
        sum = __builtin_amdgcn_sad_u16(a.x, b.v[0], sum);
        sum = __builtin_amdgcn_sad_u16(a.y, b.v[1], sum);
        sum = __builtin_amdgcn_sad_u16(a.z, b.v[2], sum);
        sum = __builtin_amdgcn_sad_u16(a.w, b.v[3], sum);
      }
    } else {
      for (int sg = q; sg < nseg; sg += 4) {
        const int yy = sg * c.row_step;
        const uint2 a = *reinterpret_cast<const uint2 *>(s_orig + yy * 4);
        const U16x4 b = *reinterpret_cast<const U16x4 *>(
            base + (size_t)(vo + (uint32_t)(yy * c.rs * 2)));
        sum = __builtin_amdgcn_sad_u16(a.x, b.v[0], sum);
        sum = __builtin_amdgcn_sad_u16(a.y, b.v[1], sum);
      }
    }
    sum += (uint32_t)lane_xor<1>((int)sum);
    sum += (uint32_t)lane_xor<2>((int)sum);
    if (q == 0) cost[i] = (sum * c.sad_mul) >> c.sad_shift;
  }
}

// ---- AC-only SAD (ComputeSadAcOnly / CalcMeanDiff, sample_metric.cc:686-703,
// :770-783): sum |a - b - avg| over the visited rows, avg = the truncated mean
// of a - b.  One candidate per LANE (the LIC instances of the kernel only: half
// of the searches of a picture that allows LIC, none of the bench's): two
// passes over the candidate's rows, the second as v_sad_u16 of the two blocks
// biased so that the mean difference cancels.
__device__ __forceinline__ uint32_t me2_sad_ac_lane(const MeCtx &c, const uint16_t *s_orig,
                                                    const uint16_t *r) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const int segw = c.w >= 8 ? 8 : 4, spr = c.w / segw, nsg = c.rows * spr;
  uint32_t sb = 0;
  for (int sg = 0; sg < nsg; sg++) {
    const int y = (sg / spr) * c.row_step, x = (sg % spr) * segw;
    const uint16_t *rp = r + (ptrdiff_t)y * c.rs + x;
    if (segw == 8) {
      const U16x8 bb = *reinterpret_cast<const U16x8 *>(rp);
#pragma unroll
      for (int k = 0; k < 4; k++) sb = __builtin_amdgcn_sad_u16(bb.v[k], 0u, sb);
    } else {
      const U16x4 bb = *reinterpret_cast<const U16x4 *>(rp);
      sb = __builtin_amdgcn_sad_u16(bb.v[0], 0u, sb);
      sb = __builtin_amdgcn_sad_u16(bb.v[1], 0u, sb);
    }
  }
  // (delta_sum * (1 + SkipLines)) / (width * height), C division
  const int num = (c.orig_sum - (int)sb) * c.sad_mul;
  const int lwh = (31 - __clz(c.w)) + (31 - __clz(c.h));
  const int avg = num >= 0 ? num >> lwh : -((-num) >> lwh);
  const uint32_t ka = (uint32_t)(8192 - avg) * 0x10001u, kb = 0x20002000u;
  auto bias = [](uint32_t v, uint32_t k) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, v) + __builtin_bit_cast(us2, k));
  };
  uint32_t sum = 0;
  for (int sg = 0; sg < nsg; sg++) {
    const int y = (sg / spr) * c.row_step, x = (sg % spr) * segw;
    const uint16_t *rp = r + (ptrdiff_t)y * c.rs + x;
    if (segw == 8) {
      const uint4 a = *reinterpret_cast<const uint4 *>(s_orig + y * c.w + x);
      const U16x8 bb = *reinterpret_cast<const U16x8 *>(rp);
      sum = __builtin_amdgcn_sad_u16(bias(a.x, ka), bias(bb.v[0], kb), sum);
      sum = __builtin_amdgcn_sad_u16(bias(a.y, ka), bias(bb.v[1], kb), sum);
      sum = __builtin_amdgcn_sad_u16(bias(a.z, ka), bias(bb.v[2], kb), sum);
      sum = __builtin_amdgcn_sad_u16(bias(a.w, ka), bias(bb.v[3], kb), sum);
    } else {
      const uint2 a = *reinterpret_cast<const uint2 *>(s_orig + y * 4);
      const U16x4 bb = *reinterpret_cast<const U16x4 *>(rp);
      sum = __builtin_amdgcn_sad_u16(bias(a.x, ka), bias(bb.v[0], kb), sum);
      sum = __builtin_amdgcn_sad_u16(bias(a.y, ka), bias(bb.v[1], kb), sum);
    }
  }
  return (sum * c.sad_mul) >> c.sad_shift;
}

__device__ __forceinline__ void me2_eval_positions_ac(const MeCtx &c, uint32_t *cost,
                                                      const uint16_t *s_orig, int n) {
  for (int i = ME2_LANE; i < n; i += 64) {
    const uint32_t pk = cost[i];
    if (pk == ME2_NOPOS) continue;
    const int x = (int)(int16_t)(pk & 0xffffu), y = (int)pk >> 16;
    cost[i] = me2_sad_ac_lane(c, s_orig, c.ref + (ptrdiff_t)y * c.rs + x);
  }
}

// Dispatch on the (wave-uniform) block shape.  Callers wave_sync() around it.
// q16: the lane's original segments of a 16-segment block (16x16, 16x8 ... whatever visits
// 16 segments), loaded once per job (me2_quad_load) - the shape of nearly every job of the
// 16 class; the other shapes fetch theirs per call.
__device__ __forceinline__ void me2_eval_positions(const MeCtx &c, uint32_t *cost,
                                                   const uint16_t *s_orig, int n,
                                                   const Me2Quad<4> &q16) {
  if (c.ac) {
    me2_eval_positions_ac(c, cost, s_orig, n);
    return;
  }
  const int nseg = c.w >= 8 ? c.rows * (c.w >> 3) : 0;
  if (nseg == 16) me2_eval_quads_reg<4>(c, cost, q16, n);
  else if (nseg == 8) {
    Me2Quad<2> qd;
    me2_quad_load<2>(c, s_orig, qd);
    me2_eval_quads_reg<2>(c, cost, qd, n);
  } else if (nseg == 4) {
    Me2Quad<1> qd;
    me2_quad_load<1>(c, s_orig, qd);
    me2_eval_quads_reg<1>(c, cost, qd, n);
  } else me2_eval_quads_gen(c, cost, s_orig, n);
}

__device__ __forceinline__ uint32_t me2_pack_pos(int x, int y) {
  return ((uint32_t)y << 16) | ((uint32_t)x & 0xffffu);
}

// This lane's two pattern entries (indices lane and lane + 64 of the TZ
// candidate pattern, tz_pattern.h), unpacked once per job.  The window tests of an
// entry (IsInside<Dir> for its one or two directions, inter_tz_search.cc:278-336) are
// kept as a packed box: the tested bounds of the job's window, the widest int16 values
// for the untested ones - an entry is inside iff clamping its packed position to the box
// leaves it unchanged (two v_pk instructions and a compare instead of a branch per
// direction).  Full-pel positions are far inside int16 (|mv| < 8192 + the range).
struct Me2Pattern {
  int meta0, meta1;  // (rng << 8) | ((d1 + d2) & 0xff), for winner look-up
  int off0, off1;    // (dy << 16) | (dx & 0xffff)
  uint32_t lo0, hi0, lo1, hi1;   // (y bound << 16) | (x bound & 0xffff)
  int round0, round1;
};

__device__ __forceinline__ void me2_pattern_box(const MeCtx &c, const TzCand &a, uint32_t &lo,
                                                uint32_t &hi) {
  const bool l = a.d1 == TZ_LEFT || a.d2 == TZ_LEFT, r = a.d1 == TZ_RIGHT || a.d2 == TZ_RIGHT;
  const bool u = a.d1 == TZ_UP || a.d2 == TZ_UP, d = a.d1 == TZ_DOWN || a.d2 == TZ_DOWN;
  lo = me2_pack_pos(l ? c.min_x : -32768, u ? c.min_y : -32768);
  hi = me2_pack_pos(r ? c.max_x : 32767, d ? c.max_y : 32767);
}

__device__ __forceinline__ Me2Pattern me2_load_pattern(const MeCtx &c, const TzCand *tzp) {
  const int lane = ME2_LANE;
  const TzCand a = tzp[lane];
  const TzCand b = tzp[lane + 64 < TZ_MAX_CANDS ? lane + 64 : 0];
  Me2Pattern p;
  p.round0 = a.round;
  p.round1 = b.round;
  p.meta0 = ((int)a.rng << 8) | ((a.d1 + a.d2) & 0xff);
  p.meta1 = ((int)b.rng << 8) | ((b.d1 + b.d2) & 0xff);
  p.off0 = ((int)a.dy << 16) | ((int)a.dx & 0xffff);
  p.off1 = ((int)b.dy << 16) | ((int)b.dx & 0xffff);
  me2_pattern_box(c, a, p.lo0, p.hi0);
  me2_pattern_box(c, b, p.lo1, p.hi1);
  return p;
}

// Pattern entry `idx` (wave-uniform) as seen from centre (bx,by).
__device__ __forceinline__ void me2_pattern_at(const Me2Pattern &p, int idx, int bx,
                                               int by, int &x, int &y, int &pos,
                                               int &rng) {
  const int l = idx & 63;
  const int o0 = __builtin_amdgcn_readlane(p.off0, l), o1 = __builtin_amdgcn_readlane(p.off1, l);
  const int m0 = __builtin_amdgcn_readlane(p.meta0, l), m1 = __builtin_amdgcn_readlane(p.meta1, l);
  const int o = (idx >> 6) ? o1 : o0, m = (idx >> 6) ? m1 : m0;
  x = bx + (int)(int16_t)(o & 0xffff);
  y = by + (o >> 16);
  pos = (int)(int8_t)(m & 0xff);
  rng = m >> 8;
}

// Evaluate pattern entries [lo, total) of the diamond list around (bx,by):
// afterwards lane l holds the keys of candidates l and l+64
// ((cost << 7) | index, or ME2_NOKEY when outside the window).
template <class SH>
__device__ __forceinline__ int me2_eval_diamonds(const MeCtx &c, SH &s,
                                                  const Me2Pattern &p, int bx, int by,
                                                  int lo, int total, uint32_t best,
                                                  uint32_t &key0, uint32_t &key1,
                                                  const Me2Quad<4> &q16) {
  const int lane = ME2_LANE;
  // packed positions of the two entries, and their window tests (Me2Pattern)
  const sp_v2s ctr = sp_s2(me2_pack_pos(bx, by));
  const sp_v2s q0 = ctr + sp_s2((uint32_t)p.off0), q1 = ctr + sp_s2((uint32_t)p.off1);
  const uint32_t pk0 = sp_u(q0), pk1 = sp_u(q1);
  const bool in0 = sp_u(__builtin_elementwise_min(__builtin_elementwise_max(q0, sp_s2(p.lo0)),
                                                  sp_s2(p.hi0))) == pk0;
  const bool in1 = sp_u(__builtin_elementwise_min(__builtin_elementwise_max(q1, sp_s2(p.lo1)),
                                                  sp_s2(p.hi1))) == pk1;
  const int x0 = (int)(int16_t)(pk0 & 0xffffu), y0 = (int)pk0 >> 16;
  const int x1 = (int)(int16_t)(pk1 & 0xffffu), y1 = (int)pk1 >> 16;
  const bool v0 = (unsigned)(lane - lo) < (unsigned)(total - lo) && in0;
  const bool v1 = (unsigned)(lane + 64 - lo) < (unsigned)(total - lo) && in1;
  // cost = dist + rate >= rate: a candidate whose rate term alone is not below
  // `best` (the running best when the sweep starts; it only decreases) can
  // never pass the strict `cost < best` test of the fold, so its SAD is not
  // needed.  Far diamonds (|mvd| >= 32 pel) mostly fall out here.
  const uint32_t r0 = (c.lambda * d_mvd_bits_fullpel(c.mvp_x, c.mvp_y, x0, y0, c.down)) >> 16;
  const uint32_t r1 = (c.lambda * d_mvd_bits_fullpel(c.mvp_x, c.mvp_y, x1, y1, c.down)) >> 16;
  const bool e0 = v0 && r0 < best, e1 = v1 && r1 < best;
  // survivors packed densely so that the 16-candidate passes stay full
  const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int n0 = __popcll(m0);
  const int i0 = __popcll(m0 & lt), i1 = n0 + __popcll(m1 & lt);
  const int n = n0 + __popcll(m1);
  wave_sync();
  if (e0) s.cost[i0] = pk0;
  if (e1) s.cost[i1] = pk1;
  wave_sync();
  me2_eval_positions(c, s.cost, s.orig, n, q16);
  wave_sync();
  key0 = key1 = ME2_NOKEY;
  if (e0) key0 = ((s.cost[i0] + r0) << 7) | (uint32_t)lane;
  if (e1) key1 = ((s.cost[i1] + r1) << 7) | (uint32_t)(lane + 64);
  return n;
}

// ---- sub-pel ---------------------------------------------------------------
// Build the horizontally filtered planes of x-phase (pel_x, fx) into slot p:
// hint = FilterHorSampleShort, hh = FilterHorSampleSample
// (inter_prediction.cc:1207-1265), all h+8 rows.
template <int MS>
__device__ __forceinline__ void me2_build_hplanes(Me2Shared<MS> &s, int bd, int w,
                                                  int h, int p, int pel_x, int fx) {
  const int lw = 31 - __clz(w);
  const int ws = w + 16;
  const int16_t *f = kLumaTaps[fx];
  const int f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4], f5 = f[5],
            f6 = f[6], f7 = f[7];
  const int shift = 6 - (14 - bd), offset = -(8192 << shift);
  const int n = (h + 8) * w;
  for (int i = ME2_LANE; i < n; i += 64) {
    const int r = i >> lw, x = i & (w - 1);
    // window column of sample (x + pel_x - 3): cols start at -8
    const uint16_t *src = s.win + r * ws + x + pel_x - 3 + 8;
    const int sum = (int)src[0] * f0 + (int)src[1] * f1 + (int)src[2] * f2 +
                    (int)src[3] * f3 + (int)src[4] * f4 + (int)src[5] * f5 +
                    (int)src[6] * f6 + (int)src[7] * f7;
    s.hint[p][i] = (int16_t)((sum + offset) >> shift);
  }
}

// One SATD pass over `ncand` candidates described in s.cand_*: lane = one tile
// row (TW samples) of one TW x TH tile of one candidate; s.dist[c] accumulates
// the normalised tile sums (ComputeSatdNxM, sample_metric.cc:403-641).
// AC: kSatdAcOnly (ComputeSatdAcOnly, sample_metric.cc:391-401): s.dsum[c] holds
// the truncated mean of orig - pred of candidate c, removed from every sample.
// SUM: no transform, only s.dsum[c] += sum(orig - pred) (the pass before).
template <int MS, int TW, int TH, bool AC = false, bool SUM = false>
__device__ __forceinline__ void me2_satd_cands(Me2Shared<MS> &s, int bd, int w,
                                               int h, int ncand) {
  const int lane = ME2_LANE;
  const int tiles_x = w / TW;
  const int upc = (w / TW) * h;  // tile rows per candidate, multiple of TH
  const int total = upc * ncand;
  const int smax = (1 << bd) - 1;
  const int16_t *lds = reinterpret_cast<const int16_t *>(s.orig);
  for (int g0 = 0; g0 < total; g0 += 64) {
    const int g = g0 + lane;
    const bool active = g < total;
    int m[TW];
    int cnd = 0;
    if (active) {
      cnd = g / upc;
      const int u = g - cnd * upc;
      const int tile = u / TH, row = u % TH;
      const int x0 = (tile % tiles_x) * TW, y = (tile / tiles_x) * TH + row;
      const int stride = s.cand_stride[cnd];
      const int16_t *pl = lds + s.cand_plane[cnd] + y * stride + x0;
      const int off = s.cand_off[cnd], sh = s.cand_shift[cnd];
      int t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = s.cand_taps[cnd][k];
      int acc[TW];
#pragma unroll
      for (int j = 0; j < TW; j++) acc[j] = off;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int16_t *rowp = pl + k * stride;
        // rows are 8-byte (TW == 4) or 16-byte aligned: packed pair loads
        uint32_t pk[TW / 2];
        if (TW == 4) {
          const uint2 v = *reinterpret_cast<const uint2 *>(rowp);
          pk[0] = v.x; pk[1] = v.y;
        } else {
#pragma unroll
          for (int q = 0; q < TW / 8; q++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(rowp + 8 * q);
            pk[4 * q + 0] = v.x; pk[4 * q + 1] = v.y; pk[4 * q + 2] = v.z; pk[4 * q + 3] = v.w;
          }
        }
#pragma unroll
        for (int j = 0; j < TW / 2; j++) {
          acc[2 * j] += (int)(int16_t)(pk[j] & 0xffff) * t[k];
          acc[2 * j + 1] += ((int)pk[j] >> 16) * t[k];
        }
      }
      const uint16_t *o = s.orig + y * w + x0;
      uint32_t ok[TW / 2];
      if (TW == 4) {
        const uint2 v = *reinterpret_cast<const uint2 *>(o);
        ok[0] = v.x; ok[1] = v.y;
      } else {
#pragma unroll
        for (int q = 0; q < TW / 8; q++) {
          const uint4 v = *reinterpret_cast<const uint4 *>(o + 8 * q);
          ok[4 * q + 0] = v.x; ok[4 * q + 1] = v.y; ok[4 * q + 2] = v.z; ok[4 * q + 3] = v.w;
        }
      }
      const int avg = AC && !SUM ? s.dsum[cnd] : 0;
#pragma unroll
      for (int j = 0; j < TW; j++) {
        const int pred = d_clip_bd((int16_t)(acc[j] >> sh), smax);
        const int ov = (j & 1) ? (int)(ok[j / 2] >> 16) : (int)(ok[j / 2] & 0xffff);
        m[j] = ov - pred - avg;
      }
    } else {
#pragma unroll
      for (int j = 0; j < TW; j++) m[j] = 0;
    }
    if (SUM) {
      int t = 0;
#pragma unroll
      for (int j = 0; j < TW; j++) t += m[j];
      t = dpp_group_sum<TH>(t);
      if (active && (lane % TH) == 0) atomicAdd(&s.dsum[cnd], t);
      continue;
    }
    // horizontal WHT in registers
#pragma unroll
    for (int len = 1; len < TW; len <<= 1)
#pragma unroll
      for (int i = 0; i < TW; i += len << 1)
#pragma unroll
        for (int j = i; j < i + len; j++) {
          const int a = m[j], b = m[j + len];
          m[j] = a + b;
          m[j + len] = a - b;
        }
    // vertical WHT across the TH lanes of the tile
    const int row = lane % TH;
    if (TH >= 2) {
      const bool up = (row & 1) != 0;
#pragma unroll
      for (int j = 0; j < TW; j++) { const int o2 = lane_xor<1>(m[j]); m[j] = up ? o2 - m[j] : m[j] + o2; }
    }
    if (TH >= 4) {
      const bool up = (row & 2) != 0;
#pragma unroll
      for (int j = 0; j < TW; j++) { const int o2 = lane_xor<2>(m[j]); m[j] = up ? o2 - m[j] : m[j] + o2; }
    }
    if (TH >= 8) {
      const bool up = (row & 4) != 0;
#pragma unroll
      for (int j = 0; j < TW; j++) { const int o2 = lane_xor<4>(m[j]); m[j] = up ? o2 - m[j] : m[j] + o2; }
    }
    if (TH >= 16) {
      const bool up = (row & 8) != 0;
#pragma unroll
      for (int j = 0; j < TW; j++) { const int o2 = lane_xor<8>(m[j]); m[j] = up ? o2 - m[j] : m[j] + o2; }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < TW; j++) sum += d_abs(m[j]);
    sum = dpp_group_sum<TH>(sum);
    if (TW == 4 && TH == 4) sum = (sum + 1) >> 1;
    else if (TW == TH) sum = (sum + 2) >> 2;
    else sum = (int)(2.0 * (double)sum / sqrt((double)(TW * TH)));
    if (active && row == 0) atomicAdd(&s.dist[cnd], (uint32_t)sum);
  }
}

template <int MS, bool AC = false, bool SUM = false>
__device__ __forceinline__ void me2_satd_dispatch(Me2Shared<MS> &s, int bd, int w,
                                                  int h, int ncand) {
  if (w == 4 && h == 4) me2_satd_cands<MS, 4, 4, AC, SUM>(s, bd, w, h, ncand);
  else if (h == 4 && w > h) me2_satd_cands<MS, 8, 4, AC, SUM>(s, bd, w, h, ncand);
  else if (w == 4 && h > w) me2_satd_cands<MS, 4, 8, AC, SUM>(s, bd, w, h, ncand);
  else if (w > h) me2_satd_cands<MS, 16, 8, AC, SUM>(s, bd, w, h, ncand);
  else if (w < h) me2_satd_cands<MS, 8, 16, AC, SUM>(s, bd, w, h, ncand);
  else me2_satd_cands<MS, 8, 8, AC, SUM>(s, bd, w, h, ncand);
}

// Offsets of the 9 half-pel / 8 quarter-pel candidates in issue order
// (kSquareXYHalf / kSquareXYQpel, inter_search.cc:38-43).
__constant__ int8_t kSubpelOff[2][9][2] = {
    {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}},
    {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}}};

// i-th candidate MV of sub-pel pass `pass` (0 = half, 1 = quarter) around base.  The
// offsets of kSubpelOff as nibbles (dx + 1) | (dy + 1) << 2 of a constant, entry i + pass: a
// shift instead of a (per-lane) table read from memory.
__device__ __forceinline__ void me2_subpel_mv(int pass, int i, int base_x, int base_y,
                                              int &mx, int &my) {
  //                   (1,1)(-1,1)(1,-1)(-1,-1)(1,0)(-1,0)(0,1)(0,-1)(0,0)
  const unsigned long long half = 0xA82064915ull;   // kSubpelOff[0][0..8]
  //                   (1,1)(-1,1)(1,0)(-1,0)(1,-1)(-1,-1)(0,1)(0,-1)
  const uint32_t quarter = 0xA8642091u;             // kSubpelOff[1][1..8]
  const uint32_t nib = pass == 0 ? (uint32_t)(half >> (4 * i)) & 15u : (quarter >> (4 * i)) & 15u;
  const int scale = pass == 0 ? 8 : 4;
  mx = base_x + ((int)(nib & 3u) - 1) * scale;
  my = base_y + ((int)(nib >> 2) - 1) * scale;
}

__device__ __forceinline__ bool me2_subpel_fast(int w, int h, int bd, bool ac = false) {
  return w >= 8 && h >= 8 && bd <= 10 && !ac;   // AC-only SATD: the 32-bit path
}

#define ME2_SLOT(MS) ((MS + 8) * (MS + 2))   // int16 entries of a plane slot

// Order in which a pass's candidates are stored for the sweep (k_subpel.h takes them two at
// a time): the pairs above / below a position first - (-1,-+1), (+1,-+1), (0,-+1) in units
// of the pass's step -, then left / right of the centre, then (half-pel pass) the centre:
// {5,7,6,8,1,2,3,4,0} / {2,6,3,7,0,1,4,5}, indices in issue order (kSubpelOff with the pass's
// skip of the centre).
__device__ __forceinline__ int me2_subpel_order(int pass, int i) {   // nibble i of a constant
  return pass == 0 ? (int)((0x043218675ull >> (4 * i)) & 15u) : (int)((0x54107362u >> (4 * i)) & 15u);
}

// Fast path of a sub-pel pass (k_subpel.h; both sides >= 8, bd <= 10, plain SATD), by one
// wave (NW = 1) or a team of NW waves: s.orig is column-major, the window is staged.
// Raw tile sums of the n = 9 - pass candidates (pass < 0: of the single MV
// (base_x, base_y)) in s.dist[i], i = the candidate's place in kSubpelOrder; returns
// this lane's candidate in issue order (lanes >= n: 64).  Arguments wave-uniform.
// held: what the four plane slots hold, a byte per slot (0x80 built, 1 from column -1 on,
// 2 two columns wider than the block), carried from pass to pass.
template <int MS, int NW>
__device__ __forceinline__ int me2_subpel_fast_pass(Me2Shared<MS> &s, const MeCtx &c,
                                                    const xvcgpu_me_block &b, int pic_w,
                                                    int pic_h, int fpx, int fpy, int pass,
                                                    int base_x, int base_y, uint32_t &held
                                                    ME2_CLK_ARG) {
  const int w = c.w, h = c.h, bd = c.bd;
  const int lane = ME2_LANE;
  const int tid = NW == 1 ? lane : (int)threadIdx.x;
  ME2_CLK_BEGIN;
  auto sync = [] {
    if (NW == 1) wave_sync();
    else __syncthreads();
  };
  const int n = pass < 0 ? 1 : 9 - pass;
  const bool need = lane < n;
  // this lane's candidate (lanes >= n idle; every wave of a team computes the same)
  const int oi = (pass >= 0 && need) ? me2_subpel_order(pass, lane) : 0;
  int mx = base_x, my = base_y;
  if (pass >= 0 && need) me2_subpel_mv(pass, oi, base_x, base_y, mx, my);
  d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);  // MotionCompensationMv, :749
  const int cpx = (mx >> 4) - fpx, cpy = (my >> 4) - fpy;   // -1 or 0 each
  const int cfx = mx & 15, cfy = my & 15;
  const int myslot = cfx >> 2;
  // per x phase in use: a plane of an earlier pass that covers the columns wanted is read
  // again, otherwise the slot is (re)built
  int build = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool mem = need && myslot == k;
    if (__ballot(mem) != 0) {
      const uint32_t m1 = __ballot(mem && cpx < 0) != 0 ? 1u : 0u;
      const uint32_t both = (m1 && __ballot(mem && cpx >= 0) != 0) ? 2u : 0u;
      const uint32_t st = (held >> (8 * k)) & 0xffu;
      const bool reuse = (st & 0x80u) && ((st & 2u) || (!both && (st & 1u) == m1));
      if (!reuse) {
        build |= 1 << k;
        held = (held & ~(0xffu << (8 * k))) | ((0x80u | m1 | both) << (8 * k));
      }
    }
  }
  sync();  // previous readers of the planes / records are done
  ME2_CLK(0);   // candidates, plane slots
  if (tid < n) {
    const int16_t *base = reinterpret_cast<const int16_t *>(s.orig);
    const int pel0 = ((held >> (8 * myslot)) & 1u) ? -1 : 0;
    const bool honly = cfx != 0 && cfy == 0, two_stage = cfx != 0 && cfy != 0;
    const int sh = 20 - bd;   // FilterVerShortSample's shift
    SpCand cd;
    cd.plane = (int)(s.hint[0] - base) + myslot * ME2_SLOT(MS) + (cpx - pel0) * (h + 8);
    cd.off = two_stage ? ((8192 << 6) + (1 << (sh - 1))) << (16 - sh)
                       : me2_honly_off(bd, cfx) << 10;
    sp_fill_taps(cd, s.taps, me2_vtaps_row(cfx, cfy), cpy + 1,
                 cfx == 0 ? 6 : (honly ? 10 : bd - 4));
    // what this candidate shares with the next one (the even lanes' records say)
    const int mine = (cfy << 4) | ((cpy + 1) << 1) | (cfy == 0 ? 1 : 0);
    const int o_mine = __builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xF, 0xF, false);
    const int o_plane = __builtin_amdgcn_update_dpp(0, cd.plane, 0xB1, 0xF, 0xF, false);
    const int o_off = __builtin_amdgcn_update_dpp(0, cd.off, 0xB1, 0xF, 0xF, false);
    int kind = SP_GENERIC;
    if (lane + 1 >= n) kind = (mine & 1) ? SP_IDENT : SP_GENERIC;   // the pair is this candidate twice
    else if ((mine & 1) && (o_mine & 1)) kind = SP_IDENT;
    else if (!((mine | o_mine) & 1) && o_plane == cd.plane && o_off == cd.off &&
             (mine >> 4) == (o_mine >> 4) && (mine & 2) == 0 && (o_mine & 2) != 0)
      kind = SP_ROWSHARED;
    sp_store_cand(s.rec + lane * SP_REC, cd, kind);
    s.dist[lane] = 0;
  }
  ME2_CLK(2);   // candidate records
#pragma unroll 1
  for (int k = 0; k < 4; k++)
    if (build & (1 << k)) {
      const uint32_t st = (held >> (8 * k)) & 0xffu;
      sp_build_planes(s.win, s.hint[0] + k * ME2_SLOT(MS), s.taps, bd, w, h, (st & 1u) ? -1 : 0,
                      4 * k, (st & 2u) ? w + 2 : w, tid, 64 * NW);
    }
  sync();
  ME2_CLK(1);   // planes
  sp_satd_pairs(reinterpret_cast<const int16_t *>(s.orig), s.rec, n, s.orig, s.dist, bd, w, h,
                tid, 64 * NW);
  sync();
  ME2_CLK(3);   // the sweep
  return need ? oi : 64;
}

// Evaluate the SATD of the n = 9 - pass candidates of a sub-pel pass (or, with
// pass < 0, of the single MV (base_x, base_y)) around the staged full-pel
// position (fpx,fpy); raw tile sums in s.dist[0..n).  Arguments wave-uniform.
// The 32-bit row-major path: 4-wide blocks, bd 12, the AC-only metric.
template <int MS>
__device__ __forceinline__ void me2_subpel_eval(Me2Shared<MS> &s, const MeCtx &c,
                                                const xvcgpu_me_block &b, int pic_w,
                                                int pic_h, int fpx, int fpy, int pass,
                                                int base_x, int base_y) {
  const int w = c.w, h = c.h, bd = c.bd;
  const int ws = w + 16;
  const int lane = ME2_LANE;
  const int n = pass < 0 ? 1 : 9 - pass;
  // this lane's candidate (lanes >= n idle)
  int mx = base_x, my = base_y;
  if (pass >= 0 && lane < n) me2_subpel_mv(pass, lane, base_x, base_y, mx, my);
  d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);  // MotionCompensationMv, :749
  const int cpx = (mx >> 4) - fpx, cpy = (my >> 4) - fpy;
  const int cfx = mx & 15, cfy = my & 15;
  // distinct horizontal phases with fx != 0 -> plane slots (<= 3)
  const int mykey = (cpx + 1) * 16 + cfx;
  const bool need = lane < n && cfx != 0;
  int myslot = -1, slot_key[3] = {-1, -1, -1};
  bool build[3] = {false, false, false};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const unsigned long long m = __ballot(need && myslot < 0);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      const int key = __builtin_amdgcn_readlane(mykey, leader);
      if (need && mykey == key) myslot = k;
      slot_key[k] = key; build[k] = true;
    }
  }
  wave_sync();  // previous readers of the planes / tables are done
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (build[k])
      me2_build_hplanes(s, bd, w, h, k, (slot_key[k] >> 4) - 1, slot_key[k] & 15);
  if (lane < n) {
    const int16_t *base = reinterpret_cast<const int16_t *>(s.orig);
    int plane, stride, off, sh;
    // plane element for output (x=0,y=0), tap 0: row (pel_y - 3) + 4
    if (cfx == 0) {
      // vertical-only / copy: FilterVerSampleSample on the window samples
      plane = (int)(reinterpret_cast<const int16_t *>(s.win) - base) +
              (cpy + 1) * ws + cpx + 8;
      stride = ws;
      off = 32;
      sh = 6;
    } else if (cfy == 0) {
      // horizontal-only: the Sample-rounded value out of the 14-bit plane
      plane = (int)(s.hint[0] - base) + myslot * ME2_SLOT(MS) + (cpy + 1) * w;
      stride = w;
      off = me2_honly_off(bd, cfx);
      sh = 6;
    } else {
      // two-stage: FilterVerShortSample on the 14-bit plane
      plane = (int)(s.hint[0] - base) + myslot * ME2_SLOT(MS) + (cpy + 1) * w;
      stride = w;
      sh = 6 + (14 - bd);
      off = (8192 << 6) + (1 << (sh - 1));
    }
    s.cand_plane[lane] = plane;
    s.cand_stride[lane] = stride;
    s.cand_off[lane] = off;
    s.cand_shift[lane] = sh;
#pragma unroll
    for (int k = 0; k < 8; k++)
      s.cand_taps[lane][k] =
          (cfx != 0 && cfy == 0) ? (k == 3 ? (int16_t)(1 << (bd - 8)) : (int16_t)0)
                                 : kLumaTaps[cfy][k];
    s.dist[lane] = 0;
    s.dsum[lane] = 0;
  }
  wave_sync();
  if (c.ac) {
    // CalcMeanDiff<0> (sample_metric.cc:770-783) per candidate, then the transform
    me2_satd_dispatch<MS, true, true>(s, bd, w, h, n);
    wave_sync();
    if (lane < n) {
      const int num = s.dsum[lane], lwh = (31 - __clz(w)) + (31 - __clz(h));
      s.dsum[lane] = num >= 0 ? num >> lwh : -((-num) >> lwh);
    }
    wave_sync();
    me2_satd_dispatch<MS, true, false>(s, bd, w, h, n);
  } else {
    me2_satd_dispatch(s, bd, w, h, n);
  }
  wave_sync();
}

// Straggler-first scheduling.  A CU whose raster hit was >= 8 away runs the
// step-5 grid: ~1500 extra candidates, ~28 us on top of a ~30 us job.  The
// launch is two rounds of waves deep, so such a job starting in the second
// round ends 20 us after everybody else - and a kernel ends with its slowest
// wave (measured with s_memrealtime, tools/trace_me.py).  Content is coherent
// in time and the job list is in picture raster order, so every job that runs
// the grid records where it sits in its XCD's share of the list, and the NEXT
// call rotates each XCD's dispatch order to start ME2_LEAD workgroups before
// the earliest recorded one: the region around last picture's slow CUs runs
// in the first round.  A rotation is a bijection of the job order: it changes
// when a job runs, never whether it runs or what it computes.
#define ME2_LEAD 64  // workgroups (~2 rows of 16x16 CUs at 1080p)
struct Me2Rot {
  int first[8];  // per XCD: workgroup (within the XCD's share) to start from
};
struct Me2Sched {
  const Me2Rot *use;  // recorded by the previous call
  Me2Rot *record;     // for the next call
  Me2Rot *clear;      // the one after that: reset here
};

// XCD-aware job index with a per-XCD rotation (see xcd_job_index).
__device__ __forceinline__ int me2_rotated_wg(int block, int n_wg, const Me2Rot *rot,
                                              int &chunk, int &local, int &len) {
  const int chunk_len = (n_wg + 7) >> 3;
  chunk = block & 7;
  const int pos = block >> 3;
  const int base = chunk * chunk_len;
  len = n_wg - base < chunk_len ? n_wg - base : chunk_len;
  if (pos >= len) return -1;  // padding workgroup (or an empty share)
  int r = rot ? rot->first[chunk] : 0;
  r = (r < 0 || r >= len) ? 0 : r;
  local = pos + r;
  local = local >= len ? local - len : local;
  return base + local;
}

// grid: ceil(n / waves) workgroups (padded to 8); block: ME2_WAVES(MS) waves, one
// job per wave.  Handles the jobs whose block fits class MS (max(w,h) <= MS)
// and no smaller class.
// PH = phases compiled in.  (Forcing more waves per SIMD onto the full-pel
// instance via a register cap spills and measured slower: 64 -> 88..149 us.)
// LIC = the instance for the jobs with XVC_ME_USE_LIC (AC-only metrics); the
// plain instances leave those jobs alone (or, when the caller did not announce
// any - lic_launched false - report them unsupported).
// One job by one wave, the descriptor read and checked (me_search_wave_body).  FW, FH > 0:
// the instance for that exact block size - the shape every loop bound, shift and tile count
// derives from is a constant then (a 1080p picture's 16x16 CUs: the bench's and most of a
// real picture's jobs); 0: any size of the class.
template <int MS, int PH, bool LIC, int FW, int FH, class Shared>
__device__ __forceinline__ void
me2_search_job(Shared &s, const PicView &orig, const PicView &ref, const xvcgpu_me_block &b_in,
               int bi, xvcgpu_me_result *results, const TzCand *tz_pattern, Me2Sched sched,
               int chunk, int local, const RefTable *refs, const uint8_t *slots) {
  constexpr bool kSched = !LIC && (PH & XVCGPU_ME_FULLPEL) != 0;
  xvcgpu_me_block b = b_in;
  if (FW > 0) {
    b.w = FW;
    b.h = FH;
  }
  const int lane = ME2_LANE;
  // (the *_refs form: the job's own reference picture out of the launch's table)
  int slot = 0;
  if (slots) {
    slot = __builtin_amdgcn_readfirstlane((int)slots[bi]);
    if (slot >= refs->n) return;
  }
  const PlaneView po = orig.c[0], pr = slots ? refs->pic[slot].c[0] : ref.c[0];
  const int pic_w = po.w, pic_h = po.h;

  MeCtx c;
  c.bd = orig.bd;
  c.w = b.w;
  c.h = b.h;
  const bool fast = b.h > 8;
  c.rows = fast ? b.h / 2 : b.h;
  c.row_step = fast ? 2 : 1;
  c.sad_mul = fast ? 2 : 1;
  c.sad_shift = c.bd - 8;
  c.rs = pr.stride;
  c.ref = pr.p + (ptrdiff_t)b.y * pr.stride + b.x;
  c.mvp_x = b.mvp_x;
  c.mvp_y = b.mvp_y;
  c.down = (b.fullpel_mv & XVC_ME_FULLPEL_MV) ? 2 : 0;
  c.lambda = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.lambda16);  // scalar: no register
  c.ac = LIC;
  c.orig_sum = 0;

  if constexpr ((PH & XVCGPU_ME_SUBPEL) != 0) {  // per-lane phase lookups come from LDS
    reinterpret_cast<uint32_t *>(&s.taps[0][0])[lane] =
        reinterpret_cast<const uint32_t *>(&kLumaTaps[0][0])[lane];
    if (lane < 8) s.taps[16][lane] = lane == 3 ? (int16_t)(1 << (c.bd - 8)) : (int16_t)0;
  }
  {  // stage the original block
    const uint16_t *o = po.p + (ptrdiff_t)b.y * po.stride + b.x;
    if (c.w >= 8) {
      wave_copy_chunks<2>(s.orig, c.w, o, po.stride, c.h, c.w >> 3);
    } else {
      for (int i = lane; i < 4 * c.h; i += 64)
        s.orig[i] = o[(ptrdiff_t)(i >> 2) * po.stride + (i & 3)];
    }
  }

  if constexpr (LIC) {  // sum of the original over the rows the full-pel metric visits
    wave_sync();
    int t = 0;
    for (int i = lane; i < c.rows * c.w; i += 64)
      t += (int)s.orig[(i / c.w) * c.row_step * c.w + (i % c.w)];
    c.orig_sum = wave_reduce_add_i32(t);
  }
  ME2_TRACE(1);  // block descriptor read, original block loads issued
  ME2_COUNT_DECL;
  xvcgpu_me_result res;
  if constexpr ((PH & XVCGPU_ME_FULLPEL) != 0) {
    const int range = b.search_range;
    d_min_max_mv(b.x, b.y, pic_w, pic_h, b.mvp_x, b.mvp_y, range, c.min_x,
                 c.min_y, c.max_x, c.max_y);
    int fs_min_x = c.min_x, fs_min_y = c.min_y, fs_max_x = c.max_x,
        fs_max_y = c.max_y;
    TzState st;
    st.bx = 0; st.by = 0; st.cost = 0xffffffffu; st.last_pos = 0; st.last_range = 0;
    Me2Quad<4> q16;
    const bool use_q16 = !LIC && c.w >= 8 && c.rows * (c.w >> 3) == 16;
    if (use_q16) {
      wave_sync();   // the staged original is visible
      me2_quad_load<4>(c, s.orig, q16);
    }

    // predictor, zero MV and previous CU's MV in one pass (groups 0..2)
    {
      int px = b.mvp_x, py = b.mvp_y;
      d_clip_mv(b.x, b.y, pic_w, pic_h, px, py);
      int qx = b.prev_x * 16, qy = b.prev_y * 16;
      d_clip_mv(b.x, b.y, pic_w, pic_h, qx, qy);
      const int ax = px >> 4, ay = py >> 4, zx = qx >> 4, zy = qy >> 4;
      uint32_t c0, c1, c2;
      if (use_q16) {
        // quads 0..2 take a candidate each: no LDS at all
        const int qi = lane >> 2;
        const int px_ = qi == 0 ? ax : (qi == 1 ? 0 : zx), py_ = qi == 0 ? ay : (qi == 1 ? 0 : zy);
        const uint32_t pk = qi < 3 ? me2_pack_pos(px_, py_) : ME2_NOPOS;
        const uint32_t cc = me_cost(c, me2_quad_sad<4>(c, q16, me2_ref_base(c), pk), px_, py_);
        c0 = (uint32_t)__builtin_amdgcn_readlane((int)cc, 0);
        c1 = (uint32_t)__builtin_amdgcn_readlane((int)cc, 4);
        c2 = (uint32_t)__builtin_amdgcn_readlane((int)cc, 8);
      } else {
        wave_sync();
        if (lane < 3)
          s.cost[lane] = me2_pack_pos(lane == 0 ? ax : (lane == 1 ? 0 : zx),
                                      lane == 0 ? ay : (lane == 1 ? 0 : zy));
        wave_sync();
        me2_eval_positions(c, s.cost, s.orig, 3, q16);
        wave_sync();
        if (lane < 3) {
          const int px_ = lane == 0 ? ax : (lane == 1 ? 0 : zx);
          const int py_ = lane == 0 ? ay : (lane == 1 ? 0 : zy);
          s.cost[lane] = me_cost(c, s.cost[lane], px_, py_);
        }
        wave_sync();
        // (the same in every lane: said so, the search state and the loops it steers stay scalar)
        c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cost[0]);
        c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cost[1]);
        c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cost[2]);
      }
      st.cost = c0; st.bx = ax; st.by = ay;
      bool change = false;
      if (st.bx != 0 || st.by != 0) {
        if (c1 < st.cost) { st.cost = c1; st.bx = 0; st.by = 0; change = true; }
      }
      st.last_range = 0;
      if (b.depth_nonzero) {
        if (c2 < st.cost) { st.cost = c2; st.bx = zx; st.by = zy; change = true; }
        if (change)
          d_min_max_mv(b.x, b.y, pic_w, pic_h, st.bx * 16, st.by * 16, range,
                       fs_min_x, fs_min_y, fs_max_x, fs_max_y);
      }
    }

    ME2_TRACE(2);  // predictor pass
    int total = 0, n_rounds = 0;
    for (int r = 1; r <= range; r *= 2) { total += tz_pattern_count(r); n_rounds++; }
    const Me2Pattern pat = me2_load_pattern(c, tz_pattern);

    // initial raster around the fixed base with per-round early termination.
    // The reference stops after 3 consecutive rounds without a hit, so at any
    // point only the next (3 - dry) rounds are certain to be examined: exactly
    // those are evaluated (one sweep), folded, and the horizon is extended only
    // if one of them hit.  The common case is ranges 1,2,4 then 8,16 - the 32
    // far candidates of ranges 32 and 64 (cold cache lines) are never read.
    {
      uint32_t k0 = ME2_NOKEY, k1 = ME2_NOKEY;
      const int bx = st.bx, by = st.by;
      int no_match = 0, r_eval = 0, idx_eval = 0;
      for (int r = 0; r < n_rounds; r++) {
        if (r >= r_eval) {
          int hi = r + (3 - no_match);
          hi = hi < n_rounds ? hi : n_rounds;
          int idx_hi = idx_eval;
          for (int q = r_eval; q < hi; q++) idx_hi += tz_pattern_count(1 << q);
          uint32_t f0, f1;
          const int ne = me2_eval_diamonds(c, s, pat, bx, by, idx_eval, idx_hi, st.cost, f0, f1, q16);
          (void)ne;
          ME2_COUNT(0, 1); ME2_COUNT(1, (ne + 15) >> 4); ME2_COUNT(4, ne);
          if (f0 != ME2_NOKEY) k0 = f0;
          if (f1 != ME2_NOKEY) k1 = f1;
          r_eval = hi;
          idx_eval = idx_hi;
        }
        uint32_t k = ME2_NOKEY;
        if (pat.round0 == r && k0 < k) k = k0;
        if (pat.round1 == r && k1 < k) k = k1;
        k = wave_min_key(k);
        bool changed = false;
        if (k != ME2_NOKEY && (k >> 7) < st.cost) {
          int x, y, pos, rng;
          me2_pattern_at(pat, (int)(k & 127), bx, by, x, y, pos, rng);
          st.cost = k >> 7; st.bx = x; st.by = y; st.last_pos = pos; st.last_range = rng;
          changed = true;
        }
        if (changed) no_match = 0;
        else if (++no_match >= 3) break;
      }
    }
    // neighbour refinement (2 candidates), shared by raster and refinement
    auto neighbor = [&]() {
      const int r = 1, bx = st.bx, by = st.by;
      int x[2], y[2], d1[2], d2[2];
      bool any = true;
      switch (st.last_pos) {
        case TZ_UP + TZ_LEFT: x[0]=bx-r;y[0]=by;d1[0]=TZ_LEFT;d2[0]=0; x[1]=bx;y[1]=by-r;d1[1]=TZ_UP;d2[1]=0; break;
        case TZ_UP: x[0]=bx-r;y[0]=by-r;d1[0]=TZ_UP;d2[0]=TZ_LEFT; x[1]=bx+r;y[1]=by-r;d1[1]=TZ_UP;d2[1]=TZ_RIGHT; break;
        case TZ_UP + TZ_RIGHT: x[0]=bx;y[0]=by-r;d1[0]=TZ_UP;d2[0]=0; x[1]=bx+r;y[1]=by;d1[1]=TZ_RIGHT;d2[1]=0; break;
        case TZ_LEFT: x[0]=bx-r;y[0]=by+r;d1[0]=TZ_DOWN;d2[0]=TZ_LEFT; x[1]=bx-r;y[1]=by-r;d1[1]=TZ_UP;d2[1]=TZ_LEFT; break;
        case TZ_RIGHT: x[0]=bx+r;y[0]=by-r;d1[0]=TZ_UP;d2[0]=TZ_RIGHT; x[1]=bx+r;y[1]=by+r;d1[1]=TZ_DOWN;d2[1]=TZ_RIGHT; break;
        case TZ_DOWN + TZ_LEFT: x[0]=bx-r;y[0]=by;d1[0]=TZ_LEFT;d2[0]=0; x[1]=bx;y[1]=by+r;d1[1]=TZ_DOWN;d2[1]=0; break;
        case TZ_DOWN: x[0]=bx-r;y[0]=by+r;d1[0]=TZ_DOWN;d2[0]=TZ_LEFT; x[1]=bx+r;y[1]=by+r;d1[1]=TZ_DOWN;d2[1]=TZ_RIGHT; break;
        case TZ_DOWN + TZ_RIGHT: x[0]=bx+r;y[0]=by;d1[0]=TZ_RIGHT;d2[0]=0; x[1]=bx;y[1]=by+r;d1[1]=TZ_DOWN;d2[1]=0; break;
        default: any = false; break;
      }
      if (!any) return;
      ME2_COUNT(3, 1);
      const bool v0 = tz_inside(c, d1[0], x[0], y[0]) && (d2[0] == 0 || tz_inside(c, d2[0], x[0], y[0]));
      const bool v1 = tz_inside(c, d1[1], x[1], y[1]) && (d2[1] == 0 || tz_inside(c, d2[1], x[1], y[1]));
      uint32_t n0, n1;
      if (use_q16) {
        const int qi = lane >> 2;
        const bool mine = qi == 0 ? v0 : (qi == 1 ? v1 : false);
        const int nx = qi == 0 ? x[0] : x[1], ny = qi == 0 ? y[0] : y[1];
        const uint32_t pk = mine ? me2_pack_pos(nx, ny) : ME2_NOPOS;
        const uint32_t cc = me_cost(c, me2_quad_sad<4>(c, q16, me2_ref_base(c), pk), nx, ny);
        n0 = (uint32_t)__builtin_amdgcn_readlane((int)cc, 0);
        n1 = (uint32_t)__builtin_amdgcn_readlane((int)cc, 4);
      } else {
        wave_sync();
        if (lane == 0) s.cost[0] = v0 ? me2_pack_pos(x[0], y[0]) : ME2_NOPOS;
        if (lane == 1) s.cost[1] = v1 ? me2_pack_pos(x[1], y[1]) : ME2_NOPOS;
        wave_sync();
        me2_eval_positions(c, s.cost, s.orig, 2, q16);
        wave_sync();
        if (lane == 0 && v0) s.cost[0] = me_cost(c, s.cost[0], x[0], y[0]);
        if (lane == 1 && v1) s.cost[1] = me_cost(c, s.cost[1], x[1], y[1]);
        wave_sync();
        n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cost[0]);
        n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cost[1]);
      }
      if (v0 && n0 < st.cost) { st.cost = n0; st.bx = x[0]; st.by = y[0]; st.last_pos = d1[0] + d2[0]; st.last_range = r; }
      if (v1 && n1 < st.cost) { st.cost = n1; st.bx = x[1]; st.by = y[1]; st.last_pos = d1[1] + d2[1]; st.last_range = r; }
    };
    ME2_TRACE(3);  // raster
    if (st.last_range == 1) { st.last_range = 0; neighbor(); }
    ME2_TRACE(4);  // neighbour
    // step-5 grid
    if (__builtin_expect(st.last_range > 5, 0)) {  // rare: spills stay inside
      if (kSched && sched.record && lane == 0)  // slow job: ask the next call to start here
        atomicMin(&sched.record->first[chunk], local > ME2_LEAD ? local - ME2_LEAD : 0);
      st.last_range = 5;
      const int nx = (fs_max_x - fs_min_x) / 5 + 1;
      const int ny = (fs_max_y - fs_min_y) / 5 + 1;
      const int tot = (fs_max_x >= fs_min_x && fs_max_y >= fs_min_y) ? nx * ny : 0;
      // Rare (a CU whose raster hit was >= 8 away) but ~1500 candidates: one
      // candidate per LANE here, so the straggler wave keeps 64 independent
      // load streams in flight instead of 4; no cross-lane reduction at all.
      uint32_t best = 0xffffffffu;
      int best_i = 0x7fffffff;
      const int segw = c.w >= 8 ? 8 : 4, spr = c.w / segw, nsg = c.rows * spr;
      for (int i = lane; i < tot; i += 64) {
        const int gx = fs_min_x + (i % nx) * 5, gy = fs_min_y + (i / nx) * 5;
        // rate-only lower bound (see me2_eval_diamonds): cannot beat the
        // running best nor this lane's own best so far
        const uint32_t rate =
            (c.lambda * d_mvd_bits_fullpel(c.mvp_x, c.mvp_y, gx, gy, c.down)) >> 16;
        if (rate >= st.cost || rate >= best) continue;
        const uint16_t *r = c.ref + (ptrdiff_t)gy * c.rs + gx;
        uint32_t sum = 0;
        if (c.ac) {
          const uint32_t cost = me_cost(c, me2_sad_ac_lane(c, s.orig, r), gx, gy);
          if (cost < best) { best = cost; best_i = i; }
          continue;
        }
        if (segw == 8 && (nsg & 7) == 0) {
          // 8 independent 16-byte loads in flight per lane
          for (int sg0 = 0; sg0 < nsg; sg0 += 8) {
            U16x8 bb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const int sg = sg0 + u;
              const int y = (sg / spr) * c.row_step, x = (sg % spr) << 3;
              bb[u] = *reinterpret_cast<const U16x8 *>(r + (ptrdiff_t)y * c.rs + x);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const int sg = sg0 + u;
              const int y = (sg / spr) * c.row_step, x = (sg % spr) << 3;
              const uint4 a = *reinterpret_cast<const uint4 *>(s.orig + y * c.w + x);
              sum = __builtin_amdgcn_sad_u16(a.x, bb[u].v[0], sum);
              sum = __builtin_amdgcn_sad_u16(a.y, bb[u].v[1], sum);
              sum = __builtin_amdgcn_sad_u16(a.z, bb[u].v[2], sum);
              sum = __builtin_amdgcn_sad_u16(a.w, bb[u].v[3], sum);
            }
          }
        } else if (segw == 8) {
          for (int sg = 0; sg < nsg; sg++) {
            const int y = (sg / spr) * c.row_step, x = (sg % spr) << 3;
            const uint4 a = *reinterpret_cast<const uint4 *>(s.orig + y * c.w + x);
            const U16x8 bb = *reinterpret_cast<const U16x8 *>(r + (ptrdiff_t)y * c.rs + x);
            sum = __builtin_amdgcn_sad_u16(a.x, bb.v[0], sum);
            sum = __builtin_amdgcn_sad_u16(a.y, bb.v[1], sum);
            sum = __builtin_amdgcn_sad_u16(a.z, bb.v[2], sum);
            sum = __builtin_amdgcn_sad_u16(a.w, bb.v[3], sum);
          }
        } else {
          for (int sg = 0; sg < nsg; sg++) {
            const int y = sg * c.row_step;
            const uint2 a = *reinterpret_cast<const uint2 *>(s.orig + y * 4);
            const U16x4 bb = *reinterpret_cast<const U16x4 *>(r + (ptrdiff_t)y * c.rs);
            sum = __builtin_amdgcn_sad_u16(a.x, bb.v[0], sum);
            sum = __builtin_amdgcn_sad_u16(a.y, bb.v[1], sum);
          }
        }
        const uint32_t cost = me_cost(c, (sum * c.sad_mul) >> c.sad_shift, gx, gy);
        if (cost < best) { best = cost; best_i = i; }
      }
      // lowest cost, then lowest index (= the reference's raster-order fold)
      const uint32_t gb = wave_min_key(best);
      const int gi = (int)wave_min_key(best == gb ? (uint32_t)best_i : 0x7fffffffu);
      if (gb < st.cost) {
        st.cost = gb;
        st.bx = fs_min_x + (gi % nx) * 5;
        st.by = fs_min_y + (gi / nx) * 5;
      }
    }
    ME2_TRACE(5);  // grid
    // iterative refinement: all diamonds around the current best, one fold
    while (st.last_range > 0) {
      st.last_range = 0;
      uint32_t k0, k1;
      const int bx = st.bx, by = st.by;
      const int ne = me2_eval_diamonds(c, s, pat, bx, by, 0, total, st.cost, k0, k1, q16);
      (void)ne;
      ME2_COUNT(0, 1); ME2_COUNT(1, (ne + 15) >> 4); ME2_COUNT(4, ne); ME2_COUNT(2, 1);
      const uint32_t k = wave_min_key(k0 < k1 ? k0 : k1);
      if (k != ME2_NOKEY && (k >> 7) < st.cost) {
        int x, y, pos, rng;
        me2_pattern_at(pat, (int)(k & 127), bx, by, x, y, pos, rng);
        st.cost = k >> 7; st.bx = x; st.by = y; st.last_pos = pos; st.last_range = rng;
      }
      if (st.last_range == 1) { st.last_range = 0; neighbor(); }
    }
    ME2_TRACE(6);  // refinement
    res.fullpel_x = st.bx;
    res.fullpel_y = st.by;
    res.fullpel_cost = st.cost;
  } else {
    res.fullpel_x = results[bi].fullpel_x;
    res.fullpel_y = results[bi].fullpel_y;
    res.fullpel_cost = results[bi].fullpel_cost;
  }
  res.mv_x = res.fullpel_x * 16;
  res.mv_y = res.fullpel_y * 16;
  res.subpel_dist = 0;

  if constexpr ((PH & XVCGPU_ME_SUBPEL) != 0) {
    const int w = c.w, h = c.h, ws = w + 16;
    const int fpx = res.fullpel_x, fpy = res.fullpel_y;
    // stage the reference window: rows -4..h+3, cols -8..w+7 around the
    // full-pel position (16-byte unaligned global loads, aligned LDS stores)
    {
      wave_sync();
      const uint16_t *r0 = pr.p + (ptrdiff_t)(b.y + fpy - 4) * pr.stride + b.x + fpx - 8;
      const int cpr = ws >> 3;  // 8-sample chunks per row (ws is 12..80: w+16)
      if ((ws & 7) == 0) {
        wave_copy_chunks<2>(s.win, ws, r0, pr.stride, h + 8, cpr);
      } else {  // w == 4: ws = 20
        for (int i = lane; i < (h + 8) * ws; i += 64) {
          const int r = i / ws, x = i - r * ws;
          s.win[i] = r0[(ptrdiff_t)r * pr.stride + x];
        }
      }
    }
    ME2_TRACE(7);  // sub-pel window loads issued
    if (me2_subpel_fast(w, h, c.bd, c.ac)) {
      // column-major original for k_subpel.h: swap across the diagonal
      wave_sync();
      const int lw = 31 - __clz(w);
      if (w == h) {
        for (int i = lane; i < w * h; i += 64) {
          const int y = i >> lw, x = i & (w - 1);
          if (x < y) {
            const uint16_t a = s.orig[i], bb = s.orig[x * w + y];
            s.orig[i] = bb;
            s.orig[x * w + y] = a;
          }
        }
      } else {  // through the (not yet built) third plane
        uint16_t *tmp = reinterpret_cast<uint16_t *>(s.hint[2]);
        for (int i = lane; i < w * h; i += 64) tmp[(i & (w - 1)) * h + (i >> lw)] = s.orig[i];
        wave_sync();
        for (int i = lane; i < w * h; i += 64) s.orig[i] = tmp[i];
      }
    }
    // one loop for the single-MV form (pass -1) and the two passes: the sweep is inlined once
    const bool fastp = me2_subpel_fast(w, h, c.bd, c.ac);
    const bool single = (b.fullpel_mv & XVC_ME_FULLPEL_MV) != 0;
    uint32_t held = 0;   // plane slots a pass leaves for the next one
    uint32_t best_cost = 0xffffffffu, best_dist = 0xffffffffu;
    int best_x = res.mv_x, best_y = res.mv_y;
    for (int pass = single ? -1 : 0; pass < (single ? 0 : 2); pass++) {
      const int base_x = best_x, base_y = best_y;
      const int n = pass < 0 ? 1 : 9 - pass;
      int oi = lane < n ? lane : 64;   // this lane's candidate, in issue order
      if (fastp)
        oi = me2_subpel_fast_pass<MS, 1>(s, c, b, pic_w, pic_h, fpx, fpy, pass, base_x, base_y,
                                         held ME2_CLK_PASS);
      else me2_subpel_eval(s, c, b, pic_w, pic_h, fpx, fpy, pass, base_x, base_y);
      if (pass < 0) {
        best_dist = s.dist[0] >> (c.bd - 8);
        break;
      }
      // the reference's ordered strict-< fold = (lowest cost, lowest index in issue order),
      // one candidate per lane
      uint32_t my_cost = 0xffffffffu, my_dist = 0;
      if (lane < n) {
        int mx, my;
        me2_subpel_mv(pass, oi, base_x, base_y, mx, my);
        my_dist = s.dist[lane] >> (c.bd - 8);
        my_cost = my_dist + ((c.lambda * d_mvd_bits(b.mvp_x, b.mvp_y, mx, my, 0)) >> 16);
      }
      const uint32_t gmin = wave_min_key(my_cost);
      const uint32_t gk = wave_min_key(my_cost == gmin ? ((uint32_t)oi << 8) | (uint32_t)lane : 0xffffu);
      if (gmin < best_cost) {
        best_cost = gmin;
        best_dist = (uint32_t)__builtin_amdgcn_readlane((int)my_dist, (int)(gk & 63u));
        me2_subpel_mv(pass, (int)(gk >> 8), base_x, base_y, best_x, best_y);
      }
    }
    res.mv_x = best_x;
    res.mv_y = best_y;
    res.subpel_dist = best_dist;
  }
  ME2_TRACE(8);  // sub-pel passes
  ME2_TRACE_RT(10);
  ME2_COUNT_STORE();
  if (lane == 0) results[bi] = res;
}

// Job bi of a search call by this wave: the descriptor read and checked, the instance
// chosen.  SEL 0: every job of the class; 1: the exact-shape jobs only - 16x16 and 16x8, the
// bottom CU row of a 1080-line picture - (me_search_sq16_kernel); 2: what that kernel
// leaves (me_search_leftover_kernel).
template <int MS, int PH, bool LIC, int SEL, typename Shared>
__device__ __forceinline__ void
me_search_wave_take(Shared &s, const PicView &orig, const PicView &ref,
                    const xvcgpu_me_block *blocks, int bi, xvcgpu_me_result *results,
                    const TzCand *tz_pattern, Me2Sched sched, int chunk, int local,
                    int max_launched, bool lic_launched, const RefTable *refs,
                    const uint8_t *slots, bool only = false) {
  ME2_TRACE(0);
  ME2_TRACE_RT(9);
  const xvcgpu_me_block b = blocks[bi];
  {
    const int mx = b.w > b.h ? b.w : b.h;
    // a job no instance of this call takes (a size the search does not have, or
    // larger than the caller's max_block_size) is answered with the
    // XVCGPU_ME_UNSUPPORTED record instead of being left as it was
    const bool pow2 = (b.w & (b.w - 1)) == 0 && (b.h & (b.h - 1)) == 0;
    const bool lic = (b.fullpel_mv & XVC_ME_USE_LIC) != 0;
    const bool valid = pow2 && b.w >= 4 && b.h >= 4 && b.w <= 64 && b.h <= 64 &&
                       mx <= max_launched && (!lic || lic_launched);
    if (lic != LIC && valid) return;  // the other set of instances
    if (MS == 16 && !LIC && !valid) {
      if (ME2_LANE == 0) {
        xvcgpu_me_result r;
        r.fullpel_x = r.fullpel_y = r.mv_x = r.mv_y = 0;
        r.fullpel_cost = r.subpel_dist = 0xffffffffu;
        results[bi] = r;
      }
      return;
    }
    if (!valid || mx > MS || (MS > 16 && mx <= MS / 2)) return;  // other class
    // me_subpel_team_kernel's jobs
    if (MS == 64 && PH == XVCGPU_ME_SUBPEL && me2_subpel_fast(b.w, b.h, orig.bd, LIC)) return;
  }
  // SEL 1: the exact-shape jobs only (me_search_sq16_kernel); SEL 2: what that kernel leaves
  const bool sq = MS == 16 && !LIC && b.w == 16 && (b.h == 16 || b.h == 8);
  if (SEL == 2 && sq) return;
  if (MS == 16 && !LIC && b.w == 16 && b.h == 16)
    me2_search_job<MS, PH, LIC, 16, 16>(s, orig, ref, b, bi, results, tz_pattern, sched, chunk, local,
                                        refs, slots);
  else if (SEL == 1 && sq)
    me2_search_job<MS, PH, LIC, 16, 8>(s, orig, ref, b, bi, results, tz_pattern, sched, chunk, local,
                                       refs, slots);
  else if (SEL != 1)
    me2_search_job<MS, PH, LIC, 0, 0>(s, orig, ref, b, bi, results, tz_pattern, sched, chunk, local,
                                      refs, slots);
  else if (only && ME2_LANE == 0) {  // XVCGPU_ME_ONLY_SQ16: nobody else takes it
    xvcgpu_me_result r;
    r.fullpel_x = r.fullpel_y = r.mv_x = r.mv_y = 0;
    r.fullpel_cost = r.subpel_dist = 0xffffffffu;
    results[bi] = r;
  }
}



template <int MS, int PH, bool LIC = false, int SEL = 0>
__device__ __forceinline__ void
me_search_wave_body(const PicView &orig, const PicView &ref,
                    const xvcgpu_me_block *blocks, int n,
                    xvcgpu_me_result *results, const TzCand *tz_pattern,
                    Me2Sched sched, int max_launched, bool lic_launched,
                    const RefTable *refs = nullptr, const uint8_t *slots = nullptr,
                    bool only = false) {
  constexpr int WPG = ME2_WAVES(MS);
  constexpr bool kSched = !LIC && (PH & XVCGPU_ME_FULLPEL) != 0;
  typedef Me2SharedT<MS, (PH & XVCGPU_ME_SUBPEL) != 0> Shared;
  __shared__ Shared s_all[WPG];
  // one wave per workgroup: the lane number is threadIdx.x itself (one register, not two)
  if (WPG == 1) __builtin_assume(threadIdx.x < 64u);
  Shared &s = s_all[threadIdx.x >> 6];
  // job = (workgroup, wave); workgroups are XCD-swizzled and rotated
  const int n_wg = (n + WPG - 1) / WPG;
  int chunk, local, len;
  const int wg = me2_rotated_wg(blockIdx.x, n_wg, kSched ? sched.use : nullptr, chunk, local,
                                len);
  if (kSched && blockIdx.x == 0 && threadIdx.x < 8) sched.clear->first[threadIdx.x] = 0x7fffffff;
  if (wg < 0) return;
  // the job index is the same in all lanes of the wave: tell the compiler, so
  // that the descriptor and everything derived from it sits in scalar registers
  // (it cost a dozen VGPRs and, under the 128-register cap, five spilled dwords)
  const int bi = __builtin_amdgcn_readfirstlane(wg * WPG + (int)(threadIdx.x >> 6));
  if (bi >= n) return;
  me_search_wave_take<MS, PH, LIC, SEL>(s, orig, ref, blocks, bi, results, tz_pattern, sched, chunk,
                                        local, max_launched, lic_launched, refs, slots, only);
}


template <int MS, int PH, bool LIC = false, int SEL = 0>
__global__ void __launch_bounds__(64 * ME2_WAVES(MS), ME2_MIN_WAVES(MS))
me_search_wave_kernel(PicView orig, PicView ref,
                      const xvcgpu_me_block *blocks, int n,
                      xvcgpu_me_result *results, const TzCand *tz_pattern,
                      Me2Sched sched, int max_launched, bool lic_launched = false) {
  me_search_wave_body<MS, PH, LIC, SEL>(orig, ref, blocks, n, results, tz_pattern, sched,
                                        max_launched, lic_launched);
}

// The exact-shape jobs of the 16 class (16x16, 16x8), both phases, in a kernel of their own
// for a job list that is (almost) all such CUs - a picture's frame pass: with the block
// size compiled in the job fits 96 registers, five waves per SIMD instead of four; the
// spilled registers sit in the step-5 grid loop, which one job in thousands runs (with
// the any-size instance in the same kernel the allocator spilled the lane number and
// the descriptor at the kernel's entry: 7.5 MB of scratch writes per 1080p launch).
// Followed by me_search_leftover_kernel (not with XVCGPU_ME_ONLY_SQ16: `only` - another shape is
// answered as unsupported then); XVCGPU_ME_HINT_SQ16 chooses the pair, callers
// with mixed sizes keep me_search_wave_kernel<16, 3>.  Measured (three pictures in
// flight): 1080p 7850 -> 8190 passes/s, 2160p 1758 -> 1828, 4320p 659 -> 695.
__global__ void __launch_bounds__(64 * ME2_WAVES(16), ME2_SQ16_MIN_WAVES)
me_search_sq16_kernel(PicView orig, PicView ref, const xvcgpu_me_block *blocks, int n,
                      xvcgpu_me_result *results, const TzCand *tz_pattern, Me2Sched sched,
                      int max_launched, bool lic_launched, bool only) {
  me_search_wave_body<16, 3, false, 1>(orig, ref, blocks, n, results, tz_pattern, sched,
                                       max_launched, lic_launched, nullptr, nullptr, only);
}

// What me_search_sq16_kernel leaves: a wave looks at 64 job descriptors, one per lane, and
// runs the ones of any other shape one after the other with the any-size instance - a
// few dozen almost empty workgroups for a frame pass's list (no such job at all on a
// 16-sample grid), correct for any list.  grid: ceil(n / 64), block 64.
__global__ void __launch_bounds__(64, ME2_MIN_WAVES(16))
me_search_leftover_kernel(PicView orig, PicView ref, const xvcgpu_me_block *blocks, int n,
                          xvcgpu_me_result *results, const TzCand *tz_pattern,
                          int max_launched, bool lic_launched) {
  typedef Me2SharedT<16, true> Shared;
  __shared__ Shared s;
  __builtin_assume(threadIdx.x < 64u);
  const int base = (int)blockIdx.x * 64, j = base + (int)threadIdx.x;
  bool mine = false;
  if (j < n) {
    const int w = blocks[j].w, h = blocks[j].h;
    mine = !(w == 16 && (h == 16 || h == 8));
  }
  unsigned long long m = __ballot(mine);
  const Me2Sched none = {nullptr, nullptr, nullptr};
  while (m) {
    const int k = __builtin_ctzll(m);
    m &= m - 1;
    const int bi = __builtin_amdgcn_readfirstlane(base + k);
    me_search_wave_take<16, 3, false, 2>(s, orig, ref, blocks, bi, results, tz_pattern, none, 0, 0,
                                         max_launched, lic_launched, nullptr, nullptr);
    wave_sync();
  }
}

// The searches of one CU state into several reference pictures in one launch: job i
// searches refs.pic[slots[i]] (a slot beyond the table: no job).  See
// xvcgpu_me_search_refs.
template <int MS, int PH>
__global__ void __launch_bounds__(64 * ME2_WAVES(MS), ME2_MIN_WAVES(MS))
me_search_refs_kernel(PicView orig, RefTable refs, const uint8_t *slots,
                      const xvcgpu_me_block *blocks, int n, xvcgpu_me_result *results,
                      const TzCand *tz_pattern, Me2Sched sched, int max_launched) {
  me_search_wave_body<MS, PH, false>(orig, orig, blocks, n, results, tz_pattern, sched,
                                     max_launched, false, &refs, slots);
}

// The searches of several pictures in one launch (grid y = picture): see
// xvcgpu_frame_pass_multi.  16-class, both phases.
struct MeMultiArgs {
  PicView orig, ref;
  const xvcgpu_me_block *blocks;
  int n;
  xvcgpu_me_result *results;
  Me2Sched sched;
};
__global__ void __launch_bounds__(64 * ME2_WAVES(16), ME2_MIN_WAVES(16))
me_search_multi_kernel(MultiArgs<MeMultiArgs> m, const TzCand *tz_pattern) {
  const MeMultiArgs &a = m.a[blockIdx.y];
  me_search_wave_body<16, 3, false>(a.orig, a.ref, a.blocks, a.n, a.results, tz_pattern, a.sched,
                                    16, false);
}

// ---- sub-pel phase of the 64 class by a team of waves -------------------------
// A 64-class job keeps ~75 KB of planes in LDS, so only two fit a CU: run by
// one wave each (the kernel above) that is two waves on four SIMDs, and the
// 64-class sub-pel phase cost 2.7x the 16 class per sample.  Here one
// workgroup of NW waves works on one job: the window, the planes and the SATD
// tiles are dealt over all its threads, the candidate set-up and the ordered
// fold are computed by every wave alike (same inputs, same result).  Takes the
// jobs of class MS that qualify for the packed path (me2_subpel_fast); the
// others stay with me_search_wave_kernel<MS, SUBPEL>, which skips these.
// grid: n workgroups padded to 8; block: 64 * NW.
template <int MS, int NW>
__device__ __forceinline__ void
me_subpel_team_body(const PicView &orig, const PicView &ref, const xvcgpu_me_block *blocks, int n,
                    xvcgpu_me_result *results, const RefTable *refs = nullptr,
                    const uint8_t *slots = nullptr) {
  __shared__ Me2Shared<MS> s;
  const int bi = xcd_job_index(blockIdx.x, n);
  if (bi < 0) return;
  const xvcgpu_me_block b = blocks[bi];
  {
    const int mx = b.w > b.h ? b.w : b.h;
    const bool pow2 = (b.w & (b.w - 1)) == 0 && (b.h & (b.h - 1)) == 0;
    if (!pow2 || mx > MS || mx <= MS / 2 ||
        !me2_subpel_fast(b.w, b.h, orig.bd, (b.fullpel_mv & XVC_ME_USE_LIC) != 0))
      return;
  }
  const int tid = threadIdx.x, lane = ME2_LANE;
  int slot = 0;
  if (slots) {
    slot = __builtin_amdgcn_readfirstlane((int)slots[bi]);
    if (slot >= refs->n) return;
  }
  const PlaneView po = orig.c[0], pr = slots ? refs->pic[slot].c[0] : ref.c[0];
  const int pic_w = po.w, pic_h = po.h;
  MeCtx c;
  c.bd = orig.bd;
  c.w = b.w;
  c.h = b.h;
  c.lambda = b.lambda16;
  c.ac = false;
  c.orig_sum = 0;
  const int w = b.w, h = b.h, ws = w + 16;
  xvcgpu_me_result res = results[bi];
  const int fpx = res.fullpel_x, fpy = res.fullpel_y;
  res.mv_x = fpx * 16;
  res.mv_y = fpy * 16;
  res.subpel_dist = 0;
  if (tid < 64)
    reinterpret_cast<uint32_t *>(&s.taps[0][0])[tid] =
        reinterpret_cast<const uint32_t *>(&kLumaTaps[0][0])[tid];
  if (tid < 8) s.taps[16][tid] = tid == 3 ? (int16_t)(1 << (orig.bd - 8)) : (int16_t)0;
  {  // the original block, column-major (column stride h) as k_subpel.h reads it
    const uint16_t *o = po.p + (ptrdiff_t)b.y * po.stride + b.x;
    const int cpr = w >> 3, lc = 31 - __clz(cpr);
    for (int i = tid; i < h * cpr; i += 64 * NW) {
      const int y = i >> lc, x0 = (i & (cpr - 1)) << 3;
      const U16x8 v = *reinterpret_cast<const U16x8 *>(o + (ptrdiff_t)y * po.stride + x0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s.orig[(x0 + 2 * k) * h + y] = (uint16_t)(v.v[k] & 0xffff);
        s.orig[(x0 + 2 * k + 1) * h + y] = (uint16_t)(v.v[k] >> 16);
      }
    }
  }
  {  // the reference window: rows -4..h+3, cols -8..w+7 around the full-pel position
    const uint16_t *r0 = pr.p + (ptrdiff_t)(b.y + fpy - 4) * pr.stride + b.x + fpx - 8;
    const int cpr = ws >> 3;
    for (int i = tid; i < (h + 8) * cpr; i += 64 * NW) {
      const int r = i / cpr, ch = i - r * cpr;
      const U16x8 v = *reinterpret_cast<const U16x8 *>(r0 + (ptrdiff_t)r * pr.stride + ch * 8);
      *reinterpret_cast<uint4 *>(s.win + r * ws + ch * 8) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    }
  }
  uint32_t held = 0;
  ME2_COUNT_DECL;
  if (b.fullpel_mv & XVC_ME_FULLPEL_MV) {
    me2_subpel_fast_pass<MS, NW>(s, c, b, pic_w, pic_h, fpx, fpy, -1, res.mv_x, res.mv_y,
                                 held ME2_CLK_PASS);
    res.subpel_dist = s.dist[0] >> (c.bd - 8);
  } else {
    uint32_t best_cost = 0xffffffffu, best_dist = 0xffffffffu;
    int best_x = res.mv_x, best_y = res.mv_y;
    for (int pass = 0; pass < 2; pass++) {
      const int base_x = best_x, base_y = best_y;
      const int nc = 9 - pass;
      const int oi = me2_subpel_fast_pass<MS, NW>(s, c, b, pic_w, pic_h, fpx, fpy, pass, base_x,
                                                  base_y, held ME2_CLK_PASS);
      uint32_t my_cost = 0xffffffffu, my_dist = 0;
      if (lane < nc) {
        int mx, my;
        me2_subpel_mv(pass, oi, base_x, base_y, mx, my);
        my_dist = s.dist[lane] >> (c.bd - 8);
        my_cost = my_dist + ((c.lambda * d_mvd_bits(b.mvp_x, b.mvp_y, mx, my, 0)) >> 16);
      }
      const uint32_t gmin = wave_min_key(my_cost);
      const uint32_t gk = wave_min_key(my_cost == gmin ? ((uint32_t)oi << 8) | (uint32_t)lane : 0xffffu);
      if (gmin < best_cost) {
        best_cost = gmin;
        best_dist = (uint32_t)__builtin_amdgcn_readlane((int)my_dist, (int)(gk & 63u));
        me2_subpel_mv(pass, (int)(gk >> 8), base_x, base_y, best_x, best_y);
      }
    }
    res.mv_x = best_x;
    res.mv_y = best_y;
    res.subpel_dist = best_dist;
  }
  if (tid == 0) results[bi] = res;
}

template <int MS, int NW>
__global__ void __launch_bounds__(64 * NW)
me_subpel_team_kernel(PicView orig, PicView ref, const xvcgpu_me_block *blocks, int n,
                      xvcgpu_me_result *results) {
  me_subpel_team_body<MS, NW>(orig, ref, blocks, n, results);
}

template <int MS, int NW>
__global__ void __launch_bounds__(64 * NW)
me_subpel_team_refs_kernel(PicView orig, RefTable refs, const uint8_t *slots,
                           const xvcgpu_me_block *blocks, int n, xvcgpu_me_result *results) {
  me_subpel_team_body<MS, NW>(orig, orig, blocks, n, results, &refs, slots);
}

#endif  // XVCGPU_K_ME2_H_
