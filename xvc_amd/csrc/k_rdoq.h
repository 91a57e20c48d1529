// k_rdoq.h -- Q2 / N2: RdoQuant::QuantRdo with CoeffSignHideRdo
// (xvc_enc_lib/rdo_quant.cc:203-446, :575-687; QuantCoeffRdo :689-720,
// EvalZeroSubblock :722-760, EvalLastPos :762-832, GetAbsLevelBits :834-878,
// UpdateCodeState :880-898, GetLastPosBits :900-947) and the context selection
// it calls (xvc_common_lib/cabac.cc:491-770, extended residual context set) -
// the quantiser the reference's encoder always runs (encoder_settings.h:59).
//
// One WAVE per transform block.  The reference walks the coefficients in
// reverse scan order, one at a time; what that order really constrains is:
//   * a coefficient's contexts read the decided levels of five neighbours to
//     its right / below (cabac.cc:535-552): up to 2 samples away, i.e. inside
//     its own 4x4 sub-block or the sub-blocks right / below / diagonal;
//   * the sub-block's coded flag context reads the flags of the sub-blocks to
//     its right and below (cabac.cc:491-518);
//   * inside a sub-block the greater1 / greater2 flag budget (c1_idx, c2_idx)
//     runs along the scan;
//   * the "last position" is the first non-zero quantised value in reverse
//     scan - a function of the input alone.
// With the extended context set nothing else crosses sub-blocks (ctx_set / c1
// only feed the non-extended contexts).  So: lane = 4x4 sub-block; the
// sub-blocks of one anti-diagonal of the sub-block grid are independent and
// run together, each lane walking its 16 coefficients in scan order - 7 steps
// for a 16x16 block instead of 16 - and every sum the reference accumulates
// along the way (int64) is formed by an order-free reduction afterwards.
// EvalLastPos is a short serial walk (it stops at the first level above 1) on
// lane 0; zero-out, re-signing and the sign-data hiding (independent per
// sub-block) are lane-parallel again.
//
// 64-point transforms only have coefficients in their 32x32 low-frequency
// corner (transform.cc:1458): the other sub-blocks matter only through the
// cost of their (zero) coded flags, which is added in closed form.
#ifndef XVCGPU_K_RDOQ_H_
#define XVCGPU_K_RDOQ_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_me.h"
#include "k_me2.h"
#include "xvcgpu_internal.h"

// ContextModel::kEntropyBits_ (context_model.cc:75-93): bits = table[state ^ bin]
__constant__ uint32_t kEntropyBits[128] = {
#include "entropy_bits.inc"
};

#define RQ_BYPASS 32768u  // ContextModel::kEntropyBypassBits
// the same table in global memory (filled by xvcgpu_create): a lookup by lane from
// __constant__ memory is one scalar load per distinct index
__device__ uint32_t gEntropyBits[128];

// Per-block scratch: N = coefficients of the (at most 32x32) low-frequency
// region; arrays indexed by rec_pos(x, y) (sub-block major, RQ_SB_STRIDE).
// Developer build (-DXVCGPU_TRACE): clock readings of the walk's sections, one
// row per workgroup of quant_rdo_packed_kernel (tools/trace_rdoq.py).
#ifdef XVCGPU_TRACE
__device__ unsigned long long g_rq_trace[4096][16];
#define RQ_TRACE(k)                                                          \
  do {                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 4096)                               \
      g_rq_trace[blockIdx.x][k] = __builtin_amdgcn_s_memtime();              \
  } while (0)
// the 100 MHz wall clock beside it (rows 2048 + workgroup, columns 0 / 1): what a
// tick of s_memtime is worth at the clock the chip runs this kernel at
#define RQ_TRACE_RT(k)                                                       \
  do {                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 2048)                               \
      g_rq_trace[2048 + blockIdx.x][k] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#else
#define RQ_TRACE(k) do {} while (0)
#define RQ_TRACE_RT(k) do {} while (0)
#endif
// step times inside the diagonal loop, summed (rows' columns 11-13, 14 = diagonals)
#ifdef XVCGPU_TRACE
#define RQ_STEP_BEGIN() unsigned long long rq_t_ = __builtin_amdgcn_s_memtime(), rq_a_[4] = {0, 0, 0, 0}
#define RQ_STEP(i)                                              \
  do {                                                          \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
    rq_a_[i] += n_ - rq_t_;                                     \
    rq_t_ = n_;                                                 \
  } while (0)
#define RQ_STEP_END()                                                           \
  do {                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 2048)                                  \
      for (int i_ = 0; i_ < 4; i_++) g_rq_trace[blockIdx.x][11 + i_] = rq_a_[i_]; \
  } while (0)
// a second set of four (EvalLastPos' parts): rows 2048 + workgroup
#define RQ_STEP2_BEGIN() rq_t_ = __builtin_amdgcn_s_memtime(); rq_a_[0] = rq_a_[1] = rq_a_[2] = rq_a_[3] = 0
#define RQ_STEP2_END()                                                          \
  do {                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 2048)                                  \
      for (int i_ = 0; i_ < 4; i_++) g_rq_trace[2048 + blockIdx.x][11 + i_] = rq_a_[i_]; \
  } while (0)
#else
#define RQ_STEP_BEGIN() do {} while (0)
#define RQ_STEP(i) do {} while (0)
#define RQ_STEP_END() do {} while (0)
#define RQ_STEP2_BEGIN() do {} while (0)
#define RQ_STEP2_END() do {} while (0)
#endif

// Per-coefficient records are laid out sub-block by sub-block with a stride of 17
// entries (rq_pos): the lanes of a group address the SAME offset of DIFFERENT
// sub-blocks together, and with the raster layout (stride 4 in x, 64 in y) the
// 16 lanes of a 16x16 block - and the four blocks of a wave - fell onto the same
// LDS banks (SQ_LDS_BANK_CONFLICT was 52 % of the LDS cycles).  17 is odd: any 16
// consecutive sub-blocks land on distinct banks for 1-, 2-, 4- and 8-byte entries.
#define RQ_SB_STRIDE 17
#define RQ_PADDED(n) ((n) + (n) / 16)
// the coefficient / level tiles of the packed kernel are copied 8 bytes at a
// time: stride 20 keeps every sub-block row 8-byte aligned
#define RQ_CF_STRIDE 20
#define RQ_CF_PADDED(n) ((n) + (n) / 4)

// wave_rdoq4's working levels (k_rdoq4.h): the corner's raster with two zero columns /
// rows to the right / below; its largest shape per coefficient budget: 4x16 / 8x32 /
// 32x32 ((w + 2) x (h + 2))
#define RQ_WL(n) ((n) + 2 * ((n) >= 1024 ? 64 : ((n) >= 256 ? 40 : 20)) + 4)

template <int N0>
struct RdoqShared {
  static constexpr int N = RQ_PADDED(N0);
  // coeff_cost_to_zero_ is not kept (8 bytes per coefficient: the records' size is
  // what limits the waves per CU): EvalLastPos recomputes a coefficient's entry
  // from what is kept - its level and its 16-bit decision record (rq ctz_of).
  // coeff_sig_bits_ / the sig-flag rate as what they are made of: the count that
  // selects the coefficient's significance context, 3 bits of its record - with
  // the records in LDS their size is what limits the waves per CU
  unsigned short rate_up[N];   // the decision-time state, 16 bits (RQ_STATE_PACK)
  // (delta_u - the quantisation error the sign hiding prices, rdo_quant.cc:360-365 - is
  // a function of the coefficient and its level: re-derived there, err_of below;
  // a sub-block's zero distortion stays in its owner lane's register)
  long long sb_code_cost[64];
  unsigned csbf_bits[64];      // csbf_bits_to_zero
  unsigned char csbf[64];
  unsigned char sb_live[64];   // the walk reaches the sub-block
  unsigned char sb_dcz[64];    // its k = 0 coefficient was decided with sig1 = 0 (rdo_quant.cc:343)
  unsigned char sb_of_scan[256];  // sub-block scan index -> sy * gw + sx
  unsigned lp_bits[32];        // EvalLastPos: last-position bits by group of x [0..15], of y [16..31]
  // GetEntropyBits(bin) of every context of the snapshot: [2 * i + bin] for the
  // context at byte offset i of xvcgpu_rdoq_contexts.  The walk looks a dozen
  // of these up per coefficient, each depending on the previous decision: from
  // global / constant memory that is a dozen dependent ~1 us round trips per
  // coefficient (the first version ran 0.86 ms per 1080p picture that way).
  alignas(8) unsigned ctx_bits[2 * sizeof(xvcgpu_rdoq_contexts)];
  alignas(8) int16_t wl[RQ_WL(N0)];
  // wave_rdoq4: the flag costs at / behind a scan position (blocks with a 64-point
  // side: up to 256 positions; the others re-use sb_code_cost)
  long long fcs_store[N0 >= 1024 ? 256 : 1];
};
template <int N0>
__device__ __forceinline__ long long *rq_fcs(RdoqShared<N0> &s) {
  return N0 >= 1024 ? s.fcs_store : s.sb_code_cost;
}

// The same members as pointers (packed kernel: per-coefficient arrays in
// global memory, the rest in LDS).
struct RdoqView {
  int16_t *wl;
  long long *fcs;
  unsigned short *rate_up;
  long long *sb_code_cost;
  unsigned *csbf_bits;
  unsigned char *csbf, *sb_live, *sb_dcz;
  unsigned char *sb_of_scan;
  unsigned *lp_bits;
  unsigned *ctx_bits;
};

// byte offsets of the context groups inside xvcgpu_rdoq_contexts
#define RQ_OFF(field) ((int)__builtin_offsetof(xvcgpu_rdoq_contexts, field))

__device__ __forceinline__ unsigned rq_bits(unsigned char state, int bin) {
  return kEntropyBits[state ^ bin];
}
__device__ __forceinline__ long long rq_bit_cost(unsigned bits, long long lambda) {
  return ((long long)bits * lambda) >> 16;
}
__device__ __forceinline__ int rq_log2(int size) {  // util::SizeToLog2, powers of two
  return 31 - __clz(size);
}
__device__ __forceinline__ int rq_last_pos_group(int pos) {  // kLastPosGroupIdx
  if (pos < 4) return pos;
  const int l = 31 - __clz(pos);                // 4..7 -> 2, 8..15 -> 3, ...
  return 2 * l + ((pos >> (l - 1)) & 1);
}
// reductions over the G (64 or 32) lanes that share a block
template <int G>
__device__ __forceinline__ int rq_wave_max_i32(int v) {
#pragma unroll
  for (int s = 1; s < G; s <<= 1) {
    const int o = __shfl_xor(v, s, 64);
    v = o > v ? o : v;
  }
  return v;
}
template <int G>
__device__ __forceinline__ long long rq_wave_sum_i64(long long v) {
#pragma unroll
  for (int s = 1; s < G; s <<= 1) v += __shfl_xor(v, s, 64);
  return v;
}
template <int G>
__device__ __forceinline__ int rq_wave_sum_i32(int v) {
#pragma unroll
  for (int s = 1; s < G; s <<= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// 64-bit sum over each aligned group of SEG (4 or 16) lanes without LDS traffic:
// three 32-bit DPP sums (the high word, and the low word as two 16-bit halves so
// that no partial sum can overflow); every lane of the group gets the total.
template <int SEG>
__device__ __forceinline__ long long rq_group_sum_i64(long long v) {
  const unsigned lo = (unsigned)v;
  const int hi = dpp_group_sum<SEG>((int)(v >> 32));
  const int l0 = dpp_group_sum<SEG>((int)(lo & 0xffffu));
  const int l1 = dpp_group_sum<SEG>((int)(lo >> 16));
  return (long long)(((unsigned long long)(unsigned)hi << 32) +
                     ((unsigned long long)(unsigned)l1 << 16) + (unsigned long long)(unsigned)l0);
}

// Inclusive prefix sum of a 64-bit value over each aligned row of 16 lanes (lane
// offsets ascending), by DPP row shifts - no LDS crossbar trips: four steps, both
// halves moved with the same control, lanes in front of the row's start read 0.
template <int N>
__device__ __forceinline__ long long rq_row_shr_add_i64(long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x110 + N, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), 0x110 + N, 0xF, 0xF, true);
  return v + (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long rq_row_scan_i64(long long v) {
  v = rq_row_shr_add_i64<1>(v);
  v = rq_row_shr_add_i64<2>(v);
  v = rq_row_shr_add_i64<4>(v);
  v = rq_row_shr_add_i64<8>(v);
  return v;
}

__device__ __forceinline__ long long *rq_fcs(RdoqView &v) { return v.fcs; }

struct RdoqCoeffState {  // RdoQuant::CoeffCodingState, the part the extended set reads
  int c1_idx, c2_idx;
  unsigned golomb_rice_k;
};

// The greater1 / greater2 flag costs of the coefficient's two contexts, loaded
// once per coefficient (c1[bin], c2[bin]); the function itself touches no memory.
struct RdoqFlagBits {
  unsigned c1_0, c1_1, c2_0, c2_1;
};
__device__ __forceinline__ unsigned rq_abs_level_bits(const RdoqFlagBits &f, int level,
                                                      const RdoqCoeffState &s) {
  const int base_level = s.c1_idx < 8 ? (2 + (s.c2_idx < 1)) : 1;
  // TransformHelper::kGolombRiceRangeExt = {6, 5, 6, 3, 3, ...} (transform.cc:61-63), as
  // arithmetic: a divergent lookup in constant memory costs a ~1 us round trip
  const unsigned threshold =
      s.golomb_rice_k < 3 ? (s.golomb_rice_k == 1 ? 5u : 6u) : 3u;
  unsigned bits = RQ_BYPASS;
  if (level >= base_level) {
    unsigned code = (unsigned)(level - base_level);
    if (code < (threshold << s.golomb_rice_k)) {
      bits += ((code >> s.golomb_rice_k) + 1 + s.golomb_rice_k) * RQ_BYPASS;
    } else {
      // the escape's prefix: the reference subtracts 2^k, 2^(k+1), ... while the
      // rest is not smaller (rdo_quant.cc:862-866); the sum of those powers is
      // 2^length - 2^k, so length = floor(log2(rest + 2^k))
      code -= threshold << s.golomb_rice_k;
      const int length = 31 - __clz((int)(code + (1u << s.golomb_rice_k)));
      bits += (unsigned)(length + (int)threshold + length + 1 - (int)s.golomb_rice_k) * RQ_BYPASS;
    }
    if (s.c1_idx < 8) {
      bits += f.c1_1;
      if (s.c2_idx < 1) bits += f.c2_1;
    }
  } else if (level == 1) {
    bits += f.c1_0;
  } else if (level == 2) {
    bits += f.c1_1 + f.c2_0;
  } else {
    return 0;
  }
  return bits;
}

// GetCoeffLastPosCtx (cabac.cc:727-770) + GetLastPosBits (rdo_quant.cc:900-947);
// returns the table index of the context
__device__ __forceinline__ int rq_last_pos_ctx(bool luma, int w, int h, int pos, bool is_x) {
  const int size = is_x ? w : h;
  if (luma) {
    const int l2 = rq_log2(size);
    const int off = l2 < 3 ? 0 : (l2 == 3 ? 3 : (l2 == 4 ? 6 : (l2 == 5 ? 10 : (l2 == 6 ? 15 : 21))));
    const int idx = off + (pos >> ((l2 + 1) >> 2));
    return 2 * ((is_x ? RQ_OFF(last_x_luma) : RQ_OFF(last_y_luma)) + idx);
  }
  const int shift = d_clip3(size >> 3, 0, 2);
  return 2 * ((is_x ? RQ_OFF(last_x_chroma) : RQ_OFF(last_y_chroma)) + (pos >> shift));
}
__device__ __forceinline__ unsigned rq_last_pos_bits(const unsigned *cb, bool luma, int w, int h,
                                                     int scan_order, int lx, int ly) {
  if (scan_order == 2) {
    int t = lx; lx = ly; ly = t;
    t = w; w = h; h = t;
  }
  const int gx = rq_last_pos_group(lx), gy = rq_last_pos_group(ly);
  unsigned bits = 0;
  int k;
  for (k = 0; k < gx; k++) bits += cb[rq_last_pos_ctx(luma, w, h, k, true) + 1];
  if (gx < rq_last_pos_group(w - 1)) bits += cb[rq_last_pos_ctx(luma, w, h, k, true)];
  for (k = 0; k < gy; k++) bits += cb[rq_last_pos_ctx(luma, w, h, k, false) + 1];
  if (gy < rq_last_pos_group(h - 1)) bits += cb[rq_last_pos_ctx(luma, w, h, k, false)];
  if (gx > 3) bits += (unsigned)((gx - 2) >> 1) * RQ_BYPASS;
  if (gy > 3) bits += (unsigned)((gy - 2) >> 1) * RQ_BYPASS;
  return bits;
}

// One axis of GetLastPosBits for a whole position GROUP g (every position of a
// group costs the same): the bits of the x (is_x) or y part in a block of
// (already scan-swapped) size w x h - rq_last_pos_bits = axis(group(x)) +
// axis(group(y)).  The context costs of all possible prefix bins are read
// together.
__device__ __forceinline__ unsigned rq_last_pos_group_bits(const unsigned *cb, bool luma, int w,
                                                           int h, int g, bool is_x) {
  const int gmax = rq_last_pos_group((is_x ? w : h) - 1);
  const int gc = gmax > 0 ? gmax - 1 : 0;
  unsigned one[10];
#pragma unroll
  for (int k = 0; k < 10; k++) one[k] = cb[rq_last_pos_ctx(luma, w, h, k < gc ? k : gc, is_x) + 1];
  const unsigned zero_bits = cb[rq_last_pos_ctx(luma, w, h, g < gc ? g : gc, is_x)];
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) bits += k < g ? one[k] : 0u;
  if (g < gmax) bits += zero_bits;
  if (g > 3) bits += (unsigned)((g - 2) >> 1) * RQ_BYPASS;
  return bits;
}

// Position of scan offset k inside a sub-block: (x, y) packed as y << 2 | x.
__device__ __forceinline__ int rq_scan_pos(int sbs, int order, int k) {
  if (sbs == 2) return (int)((d_scan4_table(order) >> (4 * k)) & 15ull);
  // TransformHelper::kScanCoeff2x2 (transform.cc:65-69): {0,2,1,3} / raster / {0,2,1,3}
  const int p = order == 1 ? k : (((k & 1) << 1) | (k >> 1));
  return ((p >> 1) << 2) | (p & 1);
}

// G lanes quantise one block: G = 64 (one block per wave), or 32 / 16 / 4 (the
// blocks of a wave side by side - `s`, cf, lev, the contexts and parameters
// are then per-lane values of the lane's own group).  cf(x, y) reads a
// transform coefficient, lev(x, y) addresses the level array (both only inside
// the region); `lane` is 0..G-1 and all G lanes must call.  Returns the number
// of non-zero levels (to every lane); levels outside the region are NOT
// written (they are zero: the caller clears what its layout needs).
// S: the scratch - RdoqShared<N> or RdoqView (the same members as pointers).
//
// What is serial and what is not.  A coefficient whose plain quantised value q
// is 0 has no choice to make (rdo_quant.cc:404-411 leaves it at level 0); it
// adds nothing to a neighbour's template nor to the sub-block's c1 / c2 budget,
// it only COSTS (its zero distortion + the bits of a zero significance flag in
// the context its decided neighbours select) and leaves the records the later
// passes read.  So a sub-block is done in three steps:
//   1. its owner lane walks the q > 0 coefficients in scan order and decides
//      them (on real content: one or two per block);
//   2. all G lanes share out the q == 0 coefficients of the sub-blocks of the
//      current anti-diagonal: costs (summed per sub-block by DPP adds) and
//      records, every template now final;
//   3. the owner evaluates the zero-sub-block choice (EvalZeroSubblock).
// The first version walked all 16 coefficients of a sub-block on its owner
// lane, ~450 instructions each at the ~5 clocks per instruction of a lone wave
// with four busy lanes: 0.1 ms for a 16x16 block and the slowest block is the
// kernel's duration.
//
// The sign hiding's rate_up / rate_down of a coefficient with a non-zero level
// (three GetAbsLevelBits evaluations) are formed when - and if - the sign
// hiding asks for them: rate_up[] then holds the coefficient's decision-time
// state (RQ_STATE_*), rate_down[] does not exist.
// 16 bits per coefficient, ALL that is kept of a decision: the template counts
// that select its three contexts - n1 / n2 = 0 for the last position, else
// min(greater-1 / greater-2 neighbours, 4) + 1 (cabac.cc:594-684), nsig =
// min(significant neighbours, 5) (cabac.cc:520-560); the contexts' position
// class is a function of x + y and is re-derived by the readers -, what
// GetAbsLevelBits reads of the budget (c1_idx < 8, c2_idx < 1) and the
// Golomb-Rice parameter (0..9).  A coefficient left at level 0 only needs n1 (the
// cost of its greater1 flag's zero bin) and nsig; RQ_STATE_NO_RATE marks the tail
// of the last sub-block, whose rate is 0.
#define RQ_STATE_PACK(n1, n2, c1_idx, c2_idx, k, nsig)                                   \
  ((unsigned short)((unsigned)(n1) | ((unsigned)(n2) << 3) |                              \
                    ((unsigned)((c1_idx) >= 8) << 6) | ((unsigned)((c2_idx) > 0) << 7) | \
                    ((unsigned)(k) << 8) | ((unsigned)(nsig) << 12)))
#define RQ_STATE_NO_RATE 0x8000u

template <int G = 64, typename S, typename CF, typename LEV>
__device__ __forceinline__ int wave_rdoq(S &s, int lane, int bd, int w, int h,
                                         int comp_qp, bool luma, int scan_order, bool sign_hide,
                                         const xvcgpu_rdoq_contexts &ctx,
                                         const xvcgpu_rdoq_params &prm, CF cf, LEV lev,
                                         bool stage_ctx = true, bool clear_levels = true) {
  const int sbs = (w == 2 || h == 2) ? 1 : 2;
  const int sb_size = 1 << (2 * sbs);
  const int gw = w >> sbs, gh = h >> sbs;                   // the whole grid (scan indices)
  const int rw = w < 32 ? w : 32, rh = h < 32 ? h : 32;     // coefficients exist here
  const int rgw = rw >> sbs, rgh = rh >> sbs;
  const int lw = rq_log2(w), lh = rq_log2(h);
  int qpb = comp_qp + 6 * (bd - 8);
  qpb = qpb > 0 ? qpb : 0;
  const int tshift = 15 - bd - ((lw + lh) >> 1);
  const bool bias = ((lw + lh) & 1) != 0;
  const int shift = 14 + qpb / 6 + tshift;
  const int size_bias_shift = bias ? 7 : 0, size_bias_offset = bias ? 64 : 0;
  const int scale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
  const int cost_scale = 15 - 2 * tshift - 2 * (bd - 8) + 2 * (bias ? 1 : 0);
  const long long lambda = prm.lambda;
  const int fq_shift = shift + (bias ? 7 : 0);
  const long long fq_offset = 1ll << (fq_shift - 1);
  const int iq_shift = 6 - tshift + (bias ? 8 : 0);
  const int iq_scale = (kInvQuantScales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);

  const bool mine = lane < rgw * rgh;
  const int sx = mine ? lane % rgw : 0, sy = mine ? lane / rgw : 0;
  const int my_scan = d_sb_scan_index(scan_order, gw, gh, sx, sy);
  const int sb_index = my_scan << (2 * sbs);
  const int px = sx << sbs, py = sy << sbs;
  auto coeff_xy = [&](int k, int &x, int &y) {
    const int p = rq_scan_pos(sbs, scan_order, k);
    x = px + (p & 3);
    y = py + (p >> 2);
  };
  // where a coefficient's records live (RQ_SB_STRIDE): sub-block major
  auto rec_pos = [&](int x, int y) {
    return sbs == 2 ? ((y >> 2) * rgw + (x >> 2)) * RQ_SB_STRIDE + (((y & 3) << 2) | (x & 3))
                    : y * rw + x;
  };
  auto quant = [&](int a) {  // GetFwdQuantFunc on a magnitude (rdo_quant.cc:949-964)
    return (int)(short)(int)((((long long)a * scale) + fq_offset) >> fq_shift);
  };
  // the template of decided neighbours (cabac.cc:535-552): five reads issued
  // together (clamped addresses, masked values) - inside the region; beyond it
  // every level is 0
  auto neighbours = [&](int x, int y, int &n_sig, int &n_g1, int &n_g2, int &sum_abs) {
    const int x1 = x + 1 < rw ? x + 1 : x, x2 = x + 2 < rw ? x + 2 : x;
    const int y1 = y + 1 < rh ? y + 1 : y, y2 = y + 2 < rh ? y + 2 : y;
    const int v0 = *lev(x1, y), v1 = *lev(x2, y), v2 = *lev(x1, y1), v3 = *lev(x, y1),
              v4 = *lev(x, y2);
    const bool m0 = x + 1 < rw, m1 = x + 2 < rw, m2 = m0 && y + 1 < rh, m3 = y + 1 < rh,
               m4 = y + 2 < rh;
    n_sig = n_g1 = n_g2 = sum_abs = 0;
    auto nb = [&](int v, bool m) {
      v = m ? d_abs(v) : 0;
      n_sig += v != 0;
      n_g1 += v > 1;
      n_g2 += v > 2;
      sum_abs += v;
    };
    nb(v0, m0); nb(v1, m1); nb(v2, m2); nb(v3, m3); nb(v4, m4);
  };
  auto sig_ctx_of = [&](int posxy, int n_sig) {  // GetCoeffSigCtx (cabac.cc:520-560)
    const int size = (lw + lh) >> 1;
    int start = posxy < 2 ? 6 : 0;
    start += luma && posxy < 5 ? 6 : 0;
    start += size > 2 && luma ? 18 << (size - 3 < 1 ? size - 3 : 1) : 0;
    const int off = n_sig < 5 ? n_sig : 5;
    return 2 * ((luma ? RQ_OFF(sig_luma) : RQ_OFF(sig_chroma)) + start + off);
  };
  // greater-1 / greater-2 context from a record's count field nn (0: last position)
  auto greater_ctx_nn = [&](int posxy, int nn) {  // cabac.cc:594-684
    const int g1 = luma ? RQ_OFF(greater1_luma) : RQ_OFF(greater1_chroma);
    const int start = luma ? (posxy < 3 ? 10 : (posxy < 10 ? 5 : 0)) : 0;
    return nn == 0 ? 2 * g1 : 2 * (g1 + start + nn);
  };
  auto greater_nn = [](int n, bool is_last) { return is_last ? 0 : (n < 4 ? n : 4) + 1; };

  // The last position: the first non-zero quantised value in reverse scan; every
  // sub-block's sum of zero costs (what a sub-block beyond the last position
  // contributes - to both totals, rdo_quant.cc:295-307 - and all it needs: its
  // levels are zero and nothing reads its per-coefficient records); and the set
  // of its coefficients that have a decision to make.
  int last = -1;
  unsigned qmask = 0;
  long long my_zero_dist = 0;
  if (mine) {
    // sum of (a * a) << cost_scale = (sum of a * a) << cost_scale: cost_scale =
    // 1 + 2 * ((lw + lh) / 2) + 2 * bias > 0, a * a < 2^31
    unsigned long long sum_sq = 0;
    for (int k = sb_size - 1; k >= 0; k--) {
      int x, y;
      coeff_xy(k, x, y);
      const int a = (short)d_abs(cf(x, y));
      sum_sq += (unsigned)(a * a);
      if (quant(a)) {
        qmask |= 1u << k;
        if (last < 0) last = sb_index + k;
      }
      if (clear_levels) *lev(x, y) = 0;   // (a caller that staged zero levels says so)
    }
    my_zero_dist = (long long)(sum_sq << cost_scale);
  }
  const int last_pos_index = rq_wave_max_i32<G>(last);
  RQ_TRACE(4);
  if (last_pos_index < 0) return 0;  // nothing quantises to a level (most blocks)

  // every context's two bin costs into LDS
  if (stage_ctx) {
    const unsigned char *cbytes = reinterpret_cast<const unsigned char *>(&ctx);
    for (int i = lane; i < (int)sizeof(xvcgpu_rdoq_contexts); i += G) {
      const unsigned char st8 = cbytes[i] & 127;
      s.ctx_bits[2 * i] = kEntropyBits[st8];
      s.ctx_bits[2 * i + 1] = kEntropyBits[st8 ^ 1];
    }
  }
  const unsigned *cb = s.ctx_bits;
  // scan index -> sub-block (whole grid), for EvalLastPos
  for (int t = lane; t < gw * gh; t += G)
    s.sb_of_scan[d_sb_scan_index(scan_order, gw, gh, t % gw, t / gw)] = (unsigned char)t;
  // a sub-block is "live" when the walk reaches it at or before the last position
  const bool live = mine && sb_index <= last_pos_index;
  // the two costs of a coefficient's significance flag as the decision used them
  // (sig1 = 0 for the last position and for the k = 0 coefficient of a sub-block
  // that was empty up to it, rdo_quant.cc:343), and its rate sig1 - sig0 (0 at and
  // beyond the last position: :303-307)
  auto sig_pair = [&](unsigned pk, int posxy, int index, int k, bool dcz, unsigned &sig0,
                      unsigned &sig1) {
    const uint2 b2 =
        *reinterpret_cast<const uint2 *>(cb + sig_ctx_of(posxy, (int)((pk >> 12) & 7u)));
    sig0 = b2.x;
    sig1 = (index == last_pos_index || (k == 0 && dcz)) ? 0u : b2.y;
  };
  auto sig_rate_of = [&](unsigned pk, int posxy, int index, int k, bool dcz) {
    if (index >= last_pos_index) return 0;
    unsigned sig0, sig1;
    sig_pair(pk, posxy, index, k, dcz, sig0, sig1);
    return (int)(sig1 - sig0);
  };
  // the greater1 / greater2 flag costs and the budget state of a decided
  // coefficient from its 16-bit record
  auto state_of = [&](unsigned pk, int posxy, RdoqFlagBits &fb, RdoqCoeffState &st) {
    const uint2 c1_b =
        *reinterpret_cast<const uint2 *>(cb + greater_ctx_nn(posxy, (int)(pk & 7u)));
    const uint2 c2_b =
        *reinterpret_cast<const uint2 *>(cb + greater_ctx_nn(posxy, (int)((pk >> 3) & 7u)));
    fb.c1_0 = c1_b.x; fb.c1_1 = c1_b.y; fb.c2_0 = c2_b.x; fb.c2_1 = c2_b.y;
    st.c1_idx = (pk >> 6) & 1u ? 8 : 0;
    st.c2_idx = (int)((pk >> 7) & 1u);
    st.golomb_rice_k = (pk >> 8) & 15u;
  };
  auto dequant = [&](int lvl) {
    int deq;
    if (iq_shift > 0) deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift;
    else deq = (lvl * iq_scale) << -iq_shift;
    return (int)(short)d_clip3(deq, -32768, 32767);
  };
  auto err_of = [&](int abs_coeff, int level) {
    const long long orig_scaled =
        (((long long)abs_coeff * scale) + size_bias_offset) >> size_bias_shift;
    const long long quant_err = orig_scaled - ((long long)level << shift);
    return (int)(short)(quant_err >> (shift - 8));
  };
  // coeff_cost_to_zero_[index] (rdo_quant.cc:350, :307) of a coefficient whose
  // sub-block is coded, recomputed: zero_cost - best_cost of the decision that
  // left it at level v.  A level of 0 - chosen, or without a choice - cost a zero
  // significance flag: -sig0.  A non-zero level cost its distortion + sig1 + the
  // level's bits in the state the decision saw.  (v < 0: a magnitude of 32768
  // wrapped, no candidate was priced and best_cost stayed at its initial value.)
  auto ctz_of = [&](unsigned pk, int posxy, int index, int k, bool dcz, int abs_coeff,
                    int v) -> long long {
    unsigned sig0, sig1;
    sig_pair(pk, posxy, index, k, dcz, sig0, sig1);
    if (v == 0) return -rq_bit_cost(sig0, lambda);
    const long long zero_cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
    if (v < 0) return zero_cost - 0x7fffffffffffffffll;
    RdoqFlagBits fb;
    RdoqCoeffState st;
    state_of(pk, posxy, fb, st);
    const unsigned bits = sig1 + rq_abs_level_bits(fb, v, st);
    const int err = abs_coeff - dequant(v);
    return zero_cost - ((((long long)err * err) << cost_scale) + rq_bit_cost(bits, lambda));
  };
  if (mine) {
    s.csbf[lane] = 0;
    s.sb_dcz[lane] = 0;
    s.sb_live[lane] = live ? 1 : 0;
    if (!live) {
      s.csbf_bits[lane] = 0;
      s.sb_code_cost[lane] = my_zero_dist;
    }
  }
  // the sub-block (region index) and offset of the last position
  const int last_k = last_pos_index & (sb_size - 1);
  const int last_l = rq_wave_max_i32<G>(mine && (last_pos_index >> (2 * sbs)) == my_scan ? lane
                                                                                          : -1);
  // the wavefront starts at the highest anti-diagonal that holds a live sub-block
  const int d_first = rq_wave_max_i32<G>(live ? sx + sy : -1);
  wave_sync();
  RQ_TRACE(5);

  // ---- one anti-diagonal of sub-blocks at a time
  RQ_STEP_BEGIN();
  for (int d = d_first; d >= 0; d--) {
    const bool act = live && sx + sy == d;
    bool any = false;
    RQ_STEP(3);
    // step 1: the owner decides the coefficients that have a choice
    if (act) {
      RdoqCoeffState st = {0, 0, 0};
      long long code_cost = 0;
      int num_non_zero = 0;
      unsigned m = qmask;
      while (m) {
        const int k = 31 - __clz((int)m);
        m ^= 1u << k;
        const int index = sb_index + k;
        int x, y;
        coeff_xy(k, x, y);
        const int pos = rec_pos(x, y);
        const int abs_coeff = (short)d_abs(cf(x, y));
        const long long zero_cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
        const int q = quant(abs_coeff);
        const bool is_last = index == last_pos_index;
        int n_sig, n_g1, n_g2, sum_abs;
        neighbours(x, y, n_sig, n_g1, n_g2, sum_abs);
        const int posxy = x + y;
        const int sig_ctx = sig_ctx_of(posxy, n_sig);
        const int nn1 = greater_nn(n_g1, is_last), nn2 = greater_nn(n_g2, is_last);
        const int c1_ctx = greater_ctx_nn(posxy, nn1);
        const int c2_ctx = greater_ctx_nn(posxy, nn2);
        {  // GetCoeffGolombRiceK (cabac.cc:686-725): smallest k with 2^(k+3) > threshold
          const unsigned threshold = 4u + (unsigned)(sum_abs - n_sig);
          const int kk = 29 - __clz((int)threshold);  // floor(log2) - 2
          st.golomb_rice_k = (unsigned)(kk < 0 ? 0 : (kk > 9 ? 9 : kk));
        }
        // the six context costs of this coefficient: three 8-byte reads, together
        const uint2 sig_b = *reinterpret_cast<const uint2 *>(cb + sig_ctx);
        const uint2 c1_b = *reinterpret_cast<const uint2 *>(cb + c1_ctx);
        const uint2 c2_b = *reinterpret_cast<const uint2 *>(cb + c2_ctx);
        const RdoqFlagBits fb = {c1_b.x, c1_b.y, c2_b.x, c2_b.y};
        const unsigned sig0 = sig_b.x;
        unsigned sig1 = sig_b.y;
        const bool dc_sig_zero = sb_index > 0 && k == 0 && num_non_zero == 0;
        if (is_last || dc_sig_zero) sig1 = 0;

        // QuantCoeffRdo (rdo_quant.cc:689-720)
        // (a magnitude of 32768 wraps to a negative q, as in the reference: no
        // candidate but zero)
        long long best_cost = 0x7fffffffffffffffll;
        int best_level = q;
        if (q > 0) {
          for (int lvl = q > 1 ? q - 1 : q; lvl <= q; lvl++) {
            const unsigned bits = sig1 + rq_abs_level_bits(fb, lvl, st);
            int deq;
            if (iq_shift > 0) deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift;
            else deq = (lvl * iq_scale) << -iq_shift;
            deq = (short)d_clip3(deq, -32768, 32767);
            const int err = abs_coeff - deq;
            const long long cost =
                (((long long)err * err) << cost_scale) + rq_bit_cost(bits, lambda);
            if (lvl == q - 1 || cost <= best_cost) {
              best_cost = cost;
              best_level = lvl;
            }
          }
        }
        if (!is_last && q < 3) {
          const long long cost = zero_cost + rq_bit_cost(sig0, lambda);
          if (cost <= best_cost) {
            best_cost = cost;
            best_level = 0;
          }
        }
        *lev(x, y) = (short)best_level;
        if (dc_sig_zero) s.sb_dcz[lane] = 1;
        code_cost += best_cost;
        if (best_level) {
          any = true;
          num_non_zero++;
        }
        s.rate_up[pos] = RQ_STATE_PACK(nn1, nn2, st.c1_idx, st.c2_idx, st.golomb_rice_k,
                                       n_sig < 5 ? n_sig : 5);
        // UpdateCodeState (rdo_quant.cc:880-898); golomb_rice_k is re-derived
        if (best_level >= 1) st.c1_idx++;
        if (best_level >= 2) st.c2_idx++;
      }
      s.sb_code_cost[lane] = code_cost;
      // EvalZeroSubblock replaces the code cost of a sub-block without a level by
      // its zero cost (:745-749; not for the first and the last sub-block, :729),
      // and nothing reads the records of such a sub-block afterwards (it is not
      // coded): its coefficients without a choice need not be priced at all.  On
      // real content that is most live sub-blocks.
      if (!any && !(sb_index == 0 || sb_index + sb_size > last_pos_index)) s.sb_live[lane] = 4;
    }
    wave_sync();
    RQ_STEP(0);
    // step 2: the coefficients without a choice, dealt over the group's lanes:
    // SEG lanes share a sub-block in every round; their costs are summed by
    // shuffles and added by the segment's first lane (16 lanes adding to one LDS
    // word with 64-bit atomics serialised)
    {
      const int ax0 = d > rgh - 1 ? d - (rgh - 1) : 0;
      const int ax1 = d < rgw - 1 ? d : rgw - 1;
      const int pairs = (ax1 - ax0 + 1) << (2 * sbs);
      constexpr int SEG = G < 16 ? G : 16;
      for (int t0 = 0; t0 < pairs; t0 += G) {
        const int t = t0 + lane;
        const bool in = t < pairs;
        const int ax = ax0 + ((in ? t : 0) >> (2 * sbs)), ay = d - ax, k = t & (sb_size - 1);
        const int l2 = ay * rgw + ax;
        long long cost = 0;
        if (in && s.sb_live[l2] == 1) {
          const int p = rq_scan_pos(sbs, scan_order, k);
          const int x = (ax << sbs) + (p & 3), y = (ay << sbs) + (p >> 2);
          const int pos = rec_pos(x, y);
          const int abs_coeff = (short)d_abs(cf(x, y));
          if (!quant(abs_coeff)) {  // (else: decided in step 1)
            cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
            if (l2 == last_l && k > last_k) {  // rdo_quant.cc:303-307 (+ the memsets :262-265)
              s.rate_up[pos] = (unsigned short)RQ_STATE_NO_RATE;
            } else {
              int n_sig, n_g1, n_g2, sum_abs;
              neighbours(x, y, n_sig, n_g1, n_g2, sum_abs);
              const int sig_ctx2 = sig_ctx_of(x + y, n_sig);
              cost += rq_bit_cost(cb[sig_ctx2], lambda);
              // (the k == 0 coefficient of an otherwise empty sub-block codes no flag,
              // sig1 = 0 at rdo_quant.cc:343; nothing reads the rate of such a sub-block)
              s.rate_up[pos] = RQ_STATE_PACK(greater_nn(n_g1, false), 0, 0, 0, 0,
                                             n_sig < 5 ? n_sig : 5);
            }
          }
        }
        // (sb_size < SEG only for 2x2 sub-blocks: then a segment is one sub-block too)
        if (sbs == 2) {
          cost = rq_group_sum_i64<SEG>(cost);
          if ((lane & (SEG - 1)) == 0 && in && cost) s.sb_code_cost[l2] += cost;
        } else {
          cost = rq_group_sum_i64<4>(cost);
          if ((lane & 3) == 0 && in && cost) s.sb_code_cost[l2] += cost;
        }
      }
    }
    wave_sync();
    RQ_STEP(1);
    // step 3: EvalZeroSubblock (rdo_quant.cc:722-760)
    bool zeroed = false;
    if (act) {
      long long sb_code_cost = s.sb_code_cost[lane];
      const long long sb_zero_dist = my_zero_dist;
      // GetSubblockCsbfCtx (cabac.cc:491-518); sub-blocks beyond the region are zero
      const bool right = sx < rgw - 1 ? s.csbf[lane + 1] != 0 : false;
      const bool below = sy < rgh - 1 ? s.csbf[lane + rgw] != 0 : false;
      const int csbf_ctx = 2 * (RQ_OFF(csbf) + (luma ? 0 : 2) + ((right || below) ? 1 : 0));
      unsigned bits_to_zero = 0;
      bool zero_sb = false;
      if (!(sb_index == 0 || sb_index + sb_size > last_pos_index)) {
        const unsigned z_bits = cb[csbf_ctx], c_bits = cb[csbf_ctx + 1];
        const long long zero_cost = sb_zero_dist + rq_bit_cost(z_bits, lambda);
        if (any) {
          const long long code_cost = sb_code_cost + rq_bit_cost(c_bits, lambda);
          if (zero_cost < code_cost) {
            sb_code_cost = zero_cost;
            bits_to_zero = z_bits;
            zero_sb = true;
          } else {
            sb_code_cost = code_cost;
            bits_to_zero = c_bits;
          }
        } else {
          sb_code_cost = zero_cost;
          bits_to_zero = z_bits;
        }
      }
      if (zero_sb) {
        any = false;
        zeroed = true;
        s.sb_live[lane] = 2;   // its levels and zero costs are cleared by all lanes below
      }
      s.csbf[lane] = any ? 1 : 0;
      s.csbf_bits[lane] = bits_to_zero;
      s.sb_code_cost[lane] = sb_code_cost;
    }
    wave_sync();
    // the sub-blocks of this diagonal that were just zeroed: their 16 levels and
    // zero costs, one coefficient per lane and step (the owner alone took 16
    // steps of two writes) - when a lane of the wave zeroed one at all
    if (__ballot(zeroed)) {
      const int ax0 = d > rgh - 1 ? d - (rgh - 1) : 0;
      const int ax1 = d < rgw - 1 ? d : rgw - 1;
      const int pairs = (ax1 - ax0 + 1) << (2 * sbs);
      for (int t = lane; t < pairs; t += G) {
        const int ax = ax0 + (t >> (2 * sbs)), ay = d - ax, k = t & (sb_size - 1);
        if (s.sb_live[ay * rgw + ax] != 2) continue;
        const int p = rq_scan_pos(sbs, scan_order, k);
        const int x = (ax << sbs) + (p & 3), y = (ay << sbs) + (p >> 2);
        *lev(x, y) = 0;
      }
      wave_sync();
    }
    RQ_STEP(2);
  }
  RQ_STEP_END();
  long long comp_code_cost = rq_wave_sum_i64<G>(mine ? s.sb_code_cost[lane] : 0ll);
  const long long comp_zero_dist = rq_wave_sum_i64<G>(mine ? my_zero_dist : 0ll);
  // sub-blocks outside the region (64-point transforms): all zero, never the
  // last one, no coded neighbour to the right / below: the cost of a zero flag
  // each (EvalZeroSubblock's csbf == 0 branch)
  const unsigned outside_bits = cb[2 * (RQ_OFF(csbf) + (luma ? 0 : 2))];
  if (gw * gh > rgw * rgh) {
    int n_out = 0;
    for (int t = lane; t < gw * gh; t += G) {
      const int tx = t % gw, ty = t / gw;
      if (tx < rgw && ty < rgh) continue;
      const int idx = d_sb_scan_index(scan_order, gw, gh, tx, ty) << (2 * sbs);
      n_out += idx > 0 && idx + sb_size <= last_pos_index;
    }
    comp_code_cost += (long long)rq_wave_sum_i32<G>(n_out) * rq_bit_cost(outside_bits, lambda);
  }

  RQ_TRACE(6);
  // ---- EvalLastPos (rdo_quant.cc:762-832).  The reference walks back from the
  // last position carrying a running cost: minus every visited sub-block's flag
  // cost, plus cost_to_zero of every visited coefficient; a non-zero level is a
  // candidate (running cost + last-position bits - its implicit sig bits), the
  // first level above 1 ends the walk.  The running cost in front of a
  // coefficient is a prefix sum in scan order: every lane sums its own
  // sub-block, takes the sum of the sub-blocks behind it (64-bit integer adds:
  // any order), evaluates its own candidates down to the stop position and the
  // group keeps the cheapest - on equal cost the one met first, i.e. the
  // highest index.  (One lane doing the walk alone cost 36 of a wave's 93 us.)
  int new_last = 0;
  const int cbf_ctx = 2 * (!luma ? RQ_OFF(cbf_chroma)
                                 : ((prm.flags & XVC_RDOQ_INTRA_CU) ? RQ_OFF(cbf_luma)
                                                                    : RQ_OFF(root_cbf)));
  const long long comp_zero_cost = comp_zero_dist + rq_bit_cost(cb[cbf_ctx], lambda);
  if (gw * gh == rgw * rgh) {
    RQ_STEP2_BEGIN();
    const int last_sb = last_pos_index >> (2 * sbs);
    const bool visited = mine && my_scan <= last_sb;
    const bool coded = visited && s.csbf[lane] != 0;
    const int start_k = my_scan == last_sb ? last_k : sb_size - 1;
    const long long flag_cost = visited ? rq_bit_cost(s.csbf_bits[lane], lambda) : 0ll;
    // the last-position bits of every position group of the two axes, once per
    // block: entries [g] for the (scan-swapped) x, [16 + g] for y - a candidate
    // then costs two reads instead of up to eighteen
    const bool lp_swap = scan_order == 2;
    constexpr int LPY = G == 4 ? 8 : 16;   // where the y groups' entries start
    {
      const int tw = lp_swap ? h : w, th = lp_swap ? w : h;
      const int nx = rq_last_pos_group(tw - 1) + 1, ny = rq_last_pos_group(th - 1) + 1;
      for (int i = lane; i < nx + ny; i += G) {
        const bool is_x = i < nx;
        const int g = is_x ? i : i - nx;
        s.lp_bits[is_x ? g : LPY + g] = rq_last_pos_group_bits(cb, luma, tw, th, g, is_x);
      }
    }
    // own sub-block, back from start_k, in ONE pass: the running sum of the
    // coefficients' cost_to_zero (recomputed, ctz_of) and, at every non-zero
    // level, the candidate's part that belongs to this sub-block (running sum +
    // last-position bits - its implicit significance bits); the cheapest is kept,
    // on equal cost the one met first.  The walk ends at the first level above 1:
    // candidates behind it - in this sub-block or in the sub-blocks below the one
    // that holds the highest such level - do not count.
    long long best_cost = 0x7fffffffffffffffll;
    int best_last_plus1 = 0;
    if (sbs == 2 && G >= 16) {
      // Sixteen lanes take the sixteen coefficients of ONE sub-block per round, from
      // the last position's sub-block down to the one that holds the highest level
      // above 1 (where the walk ends): on real content the walk passes one or two
      // coded sub-blocks, while every owner lane walking its own sixteen
      // coefficients cost sixteen serial steps whatever the content.  The running
      // cost in front of a coefficient is an exclusive prefix sum over the lanes.
      int stop_local = -1;
      if (coded) {
        // sixteen independent reads, then the highest offset with a level above 1
        // (a loop that stops at the first one waits for every read in turn)
        unsigned gt1 = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          int x, y;
          coeff_xy(k, x, y);
          gt1 |= (unsigned)((int)*lev(x, y) > 1) << k;
        }
        gt1 &= (2u << start_k) - 1u;
        if (gt1) stop_local = sb_index + 31 - __clz((int)gt1);
      }
      const int stop_idx = rq_wave_max_i32<G>(stop_local);
      const int stop_sb = stop_idx >= 0 ? stop_idx >> 4 : 0;
      RQ_STEP(0);
      const int kk = lane & 15;
      const bool worker = lane < 16;     // (G = 64: the first sixteen lanes of the wave)
      const int p = rq_scan_pos(sbs, scan_order, kk);
      const long long base = comp_code_cost + rq_bit_cost(cb[cbf_ctx + 1], lambda);
      long long acc = 0;                 // sum of (run - flag cost) of the sub-blocks behind
      for (int j = last_sb; j >= stop_sb; j--) {
        const int t = (int)s.sb_of_scan[j];
        const long long fcost = rq_bit_cost(s.csbf_bits[t], lambda);
        if (!s.csbf[t]) {                // not coded: only its flag's cost leaves the total
          acc -= fcost;
          continue;
        }
        const bool dcz = s.sb_dcz[t] != 0;
        const int x = ((t % rgw) << 2) + (p & 3), y = ((t / rgw) << 2) + (p >> 2);
        const int index = (j << 4) + kk;
        const int first_k = j == last_sb ? last_k : 15;
        const bool in = worker && kk <= first_k && index >= stop_idx;
        const unsigned pk = (unsigned)s.rate_up[rec_pos(x, y)];
        const int v = (int)*lev(x, y);
        const int ac = (short)d_abs(cf(x, y));
        unsigned sig0, sig1;
        sig_pair(pk, x + y, index, kk, dcz, sig0, sig1);
        long long ctz = 0;
        if (in && index != stop_idx) {   // (the level that ends the walk adds nothing)
          ctz = -rq_bit_cost(sig0, lambda);
          if (v == 1) {
            // GetAbsLevelBits (rdo_quant.cc:844-878) for quant_level = 1
            const unsigned c1_0 = cb[greater_ctx_nn(x + y, (int)(pk & 7u))];
            const unsigned bits1 = sig1 + (((pk >> 6) & 1u) ? (2u + ((pk >> 8) & 15u)) * RQ_BYPASS
                                                           : RQ_BYPASS + c1_0);
            const int err = ac - dequant(1);
            ctz = (((long long)(ac * ac)) << cost_scale) -
                  ((((long long)err * err) << cost_scale) + rq_bit_cost(bits1, lambda));
          } else if (v != 0) {
            ctz = ctz_of(pk, x + y, index, kk, dcz, ac, v);
          }
        }
        // inclusive prefix over the segment's lanes (scan offsets ascending), then
        // the sum of the coefficients BEHIND this one = total - inclusive
        const long long inc = rq_row_scan_i64(ctz);
        const long long total = rq_group_sum_i64<16>(ctz);
        if (in && v != 0) {
          const unsigned lp_bits = s.lp_bits[rq_last_pos_group(lp_swap ? y : x)] +
                                   s.lp_bits[LPY + rq_last_pos_group(lp_swap ? x : y)];
          const long long cost = base - fcost + acc + (total - inc) +
                                 rq_bit_cost(lp_bits, lambda) - rq_bit_cost(sig1, lambda);
          if (cost < best_cost) {        // (equal cost: the one met first, the higher index)
            best_cost = cost;
            best_last_plus1 = index + 1;
          }
        }
        acc += total - fcost;
      }
      RQ_STEP(1);
    } else {
      long long run = 0, part_best = 0x7fffffffffffffffll;
      int part_k = -1, stop_local = -1;
      if (coded) {
        const bool dcz = s.sb_dcz[lane] != 0;
        // four coefficients' reads in flight at a time (the walk itself is serial:
        // the running sum, and it ends at the first level above 1)
        for (int kb = (sb_size - 1) & ~3; kb >= 0 && stop_local < 0; kb -= 4) {
          int v[4], ac[4], xs[4], ys[4];
          unsigned pk[4], sig0[4], sig1[4];
  #pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = kb + 3 - i;
            int x, y;
            coeff_xy(k < sb_size ? k : 0, x, y);
            xs[i] = x;
            ys[i] = y;
            pk[i] = (unsigned)s.rate_up[rec_pos(x, y)];
            v[i] = (int)*lev(x, y);
            ac[i] = (short)d_abs(cf(x, y));
            sig_pair(pk[i], x + y, sb_index + k, k, dcz, sig0[i], sig1[i]);
          }
  #pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = kb + 3 - i;
            if (k >= sb_size || k > start_k || stop_local >= 0) continue;
            long long ctz = -rq_bit_cost(sig0[i], lambda);
            if (v[i]) {
              const unsigned lp_bits = s.lp_bits[rq_last_pos_group(lp_swap ? ys[i] : xs[i])] +
                                       s.lp_bits[LPY + rq_last_pos_group(lp_swap ? xs[i] : ys[i])];
              const long long part =
                  run + rq_bit_cost(lp_bits, lambda) - rq_bit_cost(sig1[i], lambda);
              if (part < part_best) {
                part_best = part;
                part_k = k;
              }
              // the walk ends at a level above 1: nothing behind it counts, its own
              // cost_to_zero included.  What it passes on its way are levels of 1 (and
              // zeros): a level of 1 costs its sign and either the zero bin of its
              // greater-1 flag or, with the flag budget spent, a Golomb-Rice code of 0
              // (GetAbsLevelBits, rdo_quant.cc:844-878, for quant_level = 1)
              if (v[i] > 1) {
                stop_local = sb_index + k;
              } else if (v[i] == 1) {
                const unsigned c1_0 = cb[greater_ctx_nn(xs[i] + ys[i], (int)(pk[i] & 7u))];
                const unsigned bits1 =
                    sig1[i] + (((pk[i] >> 6) & 1u) ? (2u + ((pk[i] >> 8) & 15u)) * RQ_BYPASS
                                                   : RQ_BYPASS + c1_0);
                const int err = ac[i] - dequant(1);
                ctz = (((long long)(ac[i] * ac[i])) << cost_scale) -
                      ((((long long)err * err) << cost_scale) + rq_bit_cost(bits1, lambda));
              } else {
                ctz = ctz_of(pk[i], xs[i] + ys[i], sb_index + k, k, dcz, ac[i], v[i]);
              }
            }
            run += ctz;
          }
        }
        // (the sum is only needed in full by the sub-blocks in FRONT of this one, and
        // they only count when no level above 1 lies here: then the loop ran to k = 0)
      }
      const long long t = run - flag_cost;
      RQ_STEP(0);
      wave_sync();   // every lane has read its sb_code_cost entry (the sums above)
      if (mine) s.sb_code_cost[my_scan] = visited ? t : 0ll;   // now indexed by scan position
      wave_sync();
      const int stop_idx = rq_wave_max_i32<G>(stop_local);
      RQ_STEP(1);
      if (part_k >= 0 && sb_index + start_k >= stop_idx && (stop_local < 0 || stop_local == stop_idx)) {
        long long c = comp_code_cost + rq_bit_cost(cb[cbf_ctx + 1], lambda) - flag_cost;
        for (int j = my_scan + 1; j <= last_sb; j++) c += s.sb_code_cost[j];
        best_cost = c + part_best;
        best_last_plus1 = sb_index + part_k + 1;
      }
    }
    RQ_STEP(2);
#pragma unroll
    for (int sh = 1; sh < G; sh <<= 1) {
      const long long oc = __shfl_xor(best_cost, sh, 64);
      const int oi = __shfl_xor(best_last_plus1, sh, 64);
      if (oc < best_cost || (oc == best_cost && oi > best_last_plus1)) {
        best_cost = oc;
        best_last_plus1 = oi;
      }
    }
    new_last = comp_zero_cost < best_cost ? -1 : best_last_plus1;
    RQ_STEP(3);
    RQ_STEP2_END();
  } else {
    // 64-point transforms (sub-blocks beyond the coefficient region take part in
    // the walk): lane 0, result broadcast
    if (lane == 0) {
      long long code_cost = comp_code_cost + rq_bit_cost(cb[cbf_ctx + 1], lambda);
      int start = last_k;
      long long best_cost = 0x7fffffffffffffffll;
      int best_last_plus1 = 0;
      bool stop = false;
      for (int sbi = last_pos_index >> (2 * sbs); sbi >= 0 && !stop; sbi--) {
        const int t = s.sb_of_scan[sbi];
        const int tx = t % gw, ty = t / gw;
        const int idx = sbi << (2 * sbs);
        if (tx >= rgw || ty >= rgh) {
          if (idx > 0 && idx + sb_size <= last_pos_index)
            code_cost -= rq_bit_cost(outside_bits, lambda);
          continue;
        }
        const int l = ty * rgw + tx;
        code_cost -= rq_bit_cost(s.csbf_bits[l], lambda);
        if (!s.csbf[l]) continue;
        for (int k = start; k >= 0; k--) {
          const int p = rq_scan_pos(sbs, scan_order, k);
          const int x = (tx << sbs) + (p & 3), y = (ty << sbs) + (p >> 2);
          const int pos = rec_pos(x, y);
          const int v = *lev(x, y);
          const unsigned pk = (unsigned)s.rate_up[pos];
          const long long ctz =
              ctz_of(pk, x + y, idx + k, k, s.sb_dcz[l] != 0, (int)(short)d_abs(cf(x, y)), v);
          if (!v) {
            code_cost += ctz;
            continue;
          }
          const unsigned lp_bits = rq_last_pos_bits(cb, luma, w, h, scan_order, x, y);
          unsigned sg0, sg1;
          sig_pair(pk, x + y, idx + k, k, s.sb_dcz[l] != 0, sg0, sg1);
          const long long cost =
              code_cost + rq_bit_cost(lp_bits, lambda) - rq_bit_cost(sg1, lambda);
          if (cost < best_cost) {
            best_cost = cost;
            best_last_plus1 = idx + k + 1;
          }
          if (v > 1) {
            stop = true;
            break;
          }
          code_cost += ctz;
        }
        start = sb_size - 1;
      }
      new_last = comp_zero_cost < best_cost ? -1 : best_last_plus1;
    }
    new_last = __shfl(new_last, (int)(ME2_LANE & ~(G - 1)), 64);
  }

  RQ_TRACE(7);
  // ---- zero what lies at / beyond the new last position, re-apply the signs
  int nnz = 0;
  bool has = false;
  if (mine)
    for (int k = 0; k < sb_size; k++) {
      int x, y;
      coeff_xy(k, x, y);
      short *out = lev(x, y);
      int level = *out;
      if (new_last < 0 || sb_index + k >= new_last) level = 0;
      nnz += level != 0;
      has |= level != 0;
      *out = (short)(cf(x, y) < 0 ? -level : level);
    }
  nnz = rq_wave_sum_i32<G>(nnz);
  if (new_last < 0) return 0;
  if (!(sign_hide && nnz > 1 && sbs > 1)) return nnz;

  RQ_TRACE(8);
  // ---- CoeffSignHideRdo (rdo_quant.cc:575-687): lane = sub-block
  const int last_sb_scan = rq_wave_max_i32<G>(has ? my_scan : -1);
  int dn = 0;
  if (mine && has) {
    const bool is_last_sb = my_scan == last_sb_scan;
    int first = 16, lastk = -1, sum = 0;
    for (int k = 15; k >= 0; k--) {
      int x, y;
      coeff_xy(k, x, y);
      const int v = *lev(x, y);
      if (v) {
        first = k < first ? k : first;
        lastk = lastk > k ? lastk : k;
        sum += v;
      }
    }
    int fx, fy;
    coeff_xy(first, fx, fy);
    const int first_sign = *lev(fx, fy) > 0 ? 0 : 1;
    if (lastk - first >= 4 && first_sign != (sum & 1)) {
      const long long rd_factor = prm.rd_factor;
      long long best_cost = 0x7fffffffffffffffll;
      int best_delta = 0, best_k = 0;
      for (int k = is_last_sb ? lastk : 15; k >= 0; k--) {
        int x, y;
        coeff_xy(k, x, y);
        const unsigned pk = (unsigned)s.rate_up[rec_pos(x, y)];
        const int lvl = *lev(x, y);
        // delta_u (rdo_quant.cc:360-365) of the level as it was decided (the sign
        // was re-applied above: undo it - a magnitude of 32768 keeps its wrapped value)
        const int coeff = cf(x, y);
        const int err_dist = err_of((short)d_abs(coeff), coeff < 0 ? -lvl : lvl);
        long long cost;
        int delta;
        if (lvl != 0) {
          // rate_up / rate_down (rdo_quant.cc:367-374) from the decision-time state
          RdoqFlagBits fb;
          RdoqCoeffState st;
          state_of(pk, x + y, fb, st);
          const int al = d_abs(lvl);
          const int lvl_rate = (int)rq_abs_level_bits(fb, al, st);
          const int rate_up = -lvl_rate + (int)rq_abs_level_bits(fb, al + 1, st);
          const int rate_down = -lvl_rate + (int)rq_abs_level_bits(fb, al - 1, st);
          const long long cost_inc = rd_factor * (-err_dist) + rate_up;
          long long cost_dec = rd_factor * err_dist + rate_down -
                               (al == 1 ? sig_rate_of(pk, x + y, sb_index + k, k, s.sb_dcz[lane] != 0) : 0);
          if (is_last_sb && k == lastk && al == 1) cost_dec -= 4ll * RQ_BYPASS;
          if (cost_inc < cost_dec) {
            cost = cost_inc;
            delta = 1;
          } else {
            delta = -1;
            cost = (k == first && al == 1) ? 0x7fffffffll : cost_dec;
          }
        } else {
          // the rate of a level that was left at 0: the zero bin of its greater1 flag
          const int rate0 =
              (pk & RQ_STATE_NO_RATE) ? 0 : (int)cb[greater_ctx_nn(x + y, (int)(pk & 7u))];
          cost = rd_factor * -(long long)d_abs(err_dist) + rate0 +
                 sig_rate_of(pk, x + y, sb_index + k, k, s.sb_dcz[lane] != 0) + (long long)RQ_BYPASS;
          delta = 1;
          if (k < first && (coeff >= 0 ? 0 : 1) != first_sign) cost = 0x7fffffffll;
        }
        if (cost < best_cost) {
          best_cost = cost;
          best_delta = delta;
          best_k = k;
        }
      }
      int x, y;
      coeff_xy(best_k, x, y);
      short *o = lev(x, y);
      const int before = *o;
      if (before == 32767 || before == -32768) best_delta = -1;
      const int after = (short)(cf(x, y) >= 0 ? before + best_delta : before - best_delta);
      *o = (short)after;
      dn = (after != 0) - (before != 0);
    }
  }
  return nnz + rq_wave_sum_i32<G>(dn);
}

// the inverse of the 4x4 scan: position y * 4 + x -> scan offset, 16 nibbles
constexpr unsigned long long rq_pack_scan4_inv(int order) {
  const unsigned long long t = tx_pack_scan4(order);
  unsigned long long r = 0;
  for (int k = 0; k < 16; k++) r |= (unsigned long long)k << (4 * (int)((t >> (4 * k)) & 15ull));
  return r;
}

// A snapshot's bit costs into cb[2 * sizeof(xvcgpu_rdoq_contexts)] by `lanes` lanes
// (lane = 0 .. lanes - 1): a word of four contexts per lane and round, then its eight
// table entries - two round trips (byte by byte: three dependent pairs).
__device__ __forceinline__ void rq_stage_costs(const xvcgpu_rdoq_contexts *snap, unsigned *cb,
                                               int lane, int lanes) {
  static_assert(sizeof(xvcgpu_rdoq_contexts) % 4 == 0, "whole words of contexts");
  constexpr int kWords = (int)sizeof(xvcgpu_rdoq_contexts) / 4;
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(snap);
  for (int wi = lane; wi < kWords; wi += lanes) {
    const uint32_t word = cw[wi];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned st8 = (word >> (8 * k)) & 127;
      *reinterpret_cast<uint2 *>(cb + 8 * wi + 2 * k) =
          make_uint2(kEntropyBits[st8], kEntropyBits[st8 ^ 1]);
    }
  }
}

#include "k_rdoq4.h"

// ---- the quantiser alone, packed ------------------------------------------------
// For flows that hold the transform coefficients (xvcgpu_fwd_transform_batch ->
// here -> xvcgpu_inv_transform_batch).  A block keeps <= 4 lanes of a wave busy
// per wavefront step, so one block per wave wastes the machine: 0.86 ms per
// 1080p picture.  Here G lanes take a block, G = its sub-block count rounded up
// to 4 / 16 / 64: a wave runs 16 blocks of up to 8x8, 4 blocks of up to 16x16 or
// one larger block, every block on its own wavefront.  A classification pass
// sorts the block indices into the three lists; everything the walk keeps per
// coefficient - the coefficient, its level and a 16-bit decision record - and the
// context costs live in LDS (9.9 KB per wave).
struct RdoqLists {
  int *list[3];       // block indices per class (4 / 16 / 64 lanes)
  int *count;         // [3]
  signed char *cls;   // per block: class, -1 = nothing to code
  int *part;          // [chunks][4]: class counts per chunk of RDOQ_CHUNK blocks (compaction)
};
#define RDOQ_CHUNK 4096   // blocks per workgroup of the compaction kernels (4 per thread)

// 0 / 1: diagonal scan, 4x4 sub-blocks, at most four / sixteen of them - four lanes
// per sub-block (wave_rdoq4); 2: everything else (the other scans, 2-wide blocks,
// more than sixteen sub-blocks, 64-point sides) - a lane per sub-block (wave_rdoq)
__device__ __forceinline__ int rq_class_of(const xvcgpu_tx_block &b) {
  if (!rq4_takes(64, b.w, b.h, (b.intra_pic >> XVC_TXF_SCAN_SHIFT) & 3) || b.w > 32 || b.h > 32)
    return 2;
  const int n_sb = (b.w >> 2) * (b.h >> 2);
  return n_sb <= 4 ? 0 : (n_sb <= 16 ? 1 : 2);
}

// Classification + the trivial case.  One wave per block: does any coefficient
// quantise to a non-zero value at all (GetFwdQuantFunc, rdo_quant.cc:949-964)?
// If not, QuantRdo returns 0 right after its scan (:387-390) - on real content
// that is most blocks (1080p QP 32: 3 of 4 luma blocks, nearly every chroma
// block): they get their zero levels and count here and never reach the serial
// walk; the others are appended to their class list.  counts must be zero on
// entry.  grid: ceil(n / 4); block: 256.
__device__ __forceinline__ void rdoq_classify_kernel_body(int bd, const xvcgpu_tx_block *blocks, int n, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, RdoqLists l) {
  const int bi = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (bi >= n) return;
  const int lane = threadIdx.x & 63;
  const xvcgpu_tx_block b = blocks[bi];
  const int w = b.w, h = b.h;
  const int lw = rq_log2(w), lh = rq_log2(h);
  int qpb = b.qp + 6 * (bd - 8);
  qpb = qpb > 0 ? qpb : 0;
  const bool bias = ((lw + lh) & 1) != 0;
  const int fq_shift = 14 + qpb / 6 + (15 - bd - ((lw + lh) >> 1)) + (bias ? 7 : 0);
  const int scale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
  const long long fq_offset = 1ll << (fq_shift - 1);
  const int16_t *src = coeffs + d_off[bi];
  int16_t *dst = levels + d_off[bi];
  bool any = false;
  for (int i = lane; i < w * h; i += 64) {
    const int a = (short)d_abs((int)src[i]);
    any |= (short)(int)((((long long)a * scale) + fq_offset) >> fq_shift) != 0;
  }
  if (__ballot(any)) {
    if (lane == 0) l.cls[bi] = (signed char)rq_class_of(b);
    return;
  }
  if (lane == 0) l.cls[bi] = -1;
  for (int i = lane; i < w * h; i += 64) dst[i] = 0;
  if (lane == 0 && nnz_out) nnz_out[bi] = 0;
}

__global__ void __launch_bounds__(256)
rdoq_classify_kernel(int bd, const xvcgpu_tx_block *blocks, int n, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, RdoqLists l) {
  rdoq_classify_kernel_body(bd, blocks, n, coeffs, d_off, levels, nnz_out, l);
}

// Blocks whose walk is bound to end in "all zero" (rdo_quant.cc:432-446: the
// coded alternative loses against cbf = 0), proved from the first pass alone and
// taken off the class lists.  With q = the plain quantised magnitude, the
// candidates for a coded level are the coefficients with q > 0 (rdo_quant.cc:
// 314-336: a coefficient with q = 0 keeps level 0).  For a last position L (a
// candidate in scan order) the coded alternative costs at least
//   cbf(1) + lastpos(L) + sum over i < L of min(zd_i, code_i) + code_last_L
//   + zd of everything else,
// where code_i >= d_best_i + lambda * (cheapest significance "1" the position can
// meet - any template count up to the number of q > 0 neighbours; none for a
// sub-block's DC behind the first sub-block, whose flag may be inferred - + sign
// + cheapest continuation: the smaller bin of a greater-1 context the position can
// meet, or a bypass bin once the budget is spent), d_best_i = the smaller of the
// distortions of q and q - 1 (the two levels the walk tries), and code_last the
// same without the significance flag.  The all-zero alternative costs cbf(0) +
// sum of zd.  So if for EVERY candidate L
//   sum_{i < L} (zd_i - min(zd_i, code_i)) + (zd_L - code_last_L)
//     < cbf(1) - cbf(0) + lastpos(L)
// then QuantRdo returns 0 for the block.  Every term is a lower bound of what the
// walk would charge, so the proof never zeroes a block the walk would code; blocks
// it cannot decide stay on the lists.  1080p QP 32, settled chain: 84 % of the luma
// blocks and nearly all chroma blocks the walk zeroes are proved here
// (tools/dbg/rdoq_zero_bound.py is the offline form of the bound).
// Blocks with 2-wide sub-blocks, a 64-point side, a magnitude of 32768, more than
// 16 candidates, coefficients that do not start on 8 bytes or another context
// snapshot than the workgroup's are left to the walk.  Sixteen lanes per block,
// RQ_PROVE_BLOCKS blocks per workgroup in rounds of sixteen (one table of context
// costs for all of them); grid: ceil(n / RQ_PROVE_BLOCKS); block: 256.
#define RQ_PROVE_BLOCKS 16
#define RQ_PROVE_MAX_CANDS 16

// a block's quantiser constants, as QuantRdo derives them (rdo_quant.cc:223-300)
struct RqProveBlock {
  int w, h, lw, lh, scale, fq_shift, cost_scale, iq_shift, iq_scale, scan_order;
  long long fq_offset, lambda;
  bool luma, intra_cu;
};

__device__ __forceinline__ RqProveBlock rq_prove_block(const xvcgpu_tx_block &b, int bd,
                                                       const xvcgpu_rdoq_params &prm) {
  RqProveBlock k;
  k.w = b.w;
  k.h = b.h;
  k.luma = b.comp == 0;
  k.intra_cu = (prm.flags & XVC_RDOQ_INTRA_CU) != 0;
  k.scan_order = (b.intra_pic >> XVC_TXF_SCAN_SHIFT) & 3;
  k.lw = rq_log2(k.w);
  k.lh = rq_log2(k.h);
  int qpb = b.qp + 6 * (bd - 8);
  qpb = qpb > 0 ? qpb : 0;
  const int tshift = 15 - bd - ((k.lw + k.lh) >> 1);
  const bool bias = ((k.lw + k.lh) & 1) != 0;
  k.scale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
  k.fq_shift = 14 + qpb / 6 + tshift + (bias ? 7 : 0);
  k.fq_offset = 1ll << (k.fq_shift - 1);
  k.cost_scale = 15 - 2 * tshift - 2 * (bd - 8) + 2 * (bias ? 1 : 0);
  k.iq_shift = 6 - tshift + (bias ? 8 : 0);
  k.iq_scale = (kInvQuantScales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
  k.lambda = prm.lambda;
  return k;
}

__device__ __forceinline__ int rq_prove_quant(const RqProveBlock &k, int a) {
  return (int)(short)(int)((((long long)a * k.scale) + k.fq_offset) >> k.fq_shift);
}

// the smallest magnitude that quantises to a level: a * scale >= 2^(shift - 1)
__device__ __forceinline__ int rq_prove_threshold(const RqProveBlock &k) {
  long long t = (long long)((double)k.fq_offset / (double)k.scale);
  while (t * k.scale < k.fq_offset) t++;
  while (t > 0 && (t - 1) * k.scale >= k.fq_offset) t--;
  return t > 32768 ? 32768 : (int)t;
}

// One candidate (x, y) with magnitude a: what coding it can save at most as an
// inner coefficient (gain) and as the last one (gain_last), what a last position
// there costs on top (rhs), and its place in the scan (idx).  qat(x, y) = the
// plain quantised magnitude at a position, 0 outside the block; cb = the
// snapshot's bit costs (entry 2 * ctx + bin).
template <class QAt>
__device__ __forceinline__ void rq_prove_candidate(const RqProveBlock &k, const unsigned *cb,
                                                   int x, int y, int a, QAt qat,
                                                   long long &gain, long long &gain_last,
                                                   long long &rhs, int &idx) {
  const int w = k.w, h = k.h;
  const bool luma = k.luma;
  const int q = rq_prove_quant(k, a);
  int cnt, cnt1;
  {
    const int q0 = qat(x + 1, y), q1 = qat(x + 2, y), q2 = qat(x + 1, y + 1), q3 = qat(x, y + 1),
              q4 = qat(x, y + 2);
    cnt = (q0 > 0) + (q1 > 0) + (q2 > 0) + (q3 > 0) + (q4 > 0);
    cnt1 = (q0 > 1) + (q1 > 1) + (q2 > 1) + (q3 > 1) + (q4 > 1);
  }
  const int posxy = x + y, size = (k.lw + k.lh) >> 1;
  // GetCoeffSigCtx (cabac.cc:520-560): the cheapest "1" bin the flag can meet
  int start = posxy < 2 ? 6 : 0;
  start += luma && posxy < 5 ? 6 : 0;
  start += size > 2 && luma ? 18 << (size - 3 < 1 ? size - 3 : 1) : 0;
  const int sig_base = (luma ? RQ_OFF(sig_luma) : RQ_OFF(sig_chroma)) + start;
  unsigned sig1 = 0xffffffffu;
  for (int nn = 0; nn <= cnt; nn++) {
    const unsigned v = cb[2 * (sig_base + nn) + 1];
    sig1 = v < sig1 ? v : sig1;
  }
  const bool sub_dc = ((x | y) & 3) == 0 && (x | y) != 0;   // k = 0 behind the first sub-block
  if (sub_dc) sig1 = 0;
  // the greater-1 contexts (cabac.cc:594-684): the last position's, or by the count
  const int g1 = luma ? RQ_OFF(greater1_luma) : RQ_OFF(greater1_chroma);
  const int gstart = luma ? (posxy < 3 ? 10 : (posxy < 10 ? 5 : 0)) : 0;
  unsigned flag_min = RQ_BYPASS;   // (budget spent: a Golomb-Rice code of at least one bin)
  {
    const unsigned v0 = cb[2 * g1], v1 = cb[2 * g1 + 1];
    flag_min = v0 < flag_min ? v0 : flag_min;
    flag_min = v1 < flag_min ? v1 : flag_min;
  }
  for (int nn = 0; nn <= cnt1; nn++) {
    const int c = 2 * (g1 + gstart + (nn < 4 ? nn : 4) + 1);
    const unsigned v0 = cb[c], v1 = cb[c + 1];
    flag_min = v0 < flag_min ? v0 : flag_min;
    flag_min = v1 < flag_min ? v1 : flag_min;
  }
  const unsigned lvl_min = RQ_BYPASS + flag_min;   // sign + the cheapest continuation
  auto dist_of = [&](int lvl) {
    int deq;
    if (k.iq_shift > 0) deq = (lvl * k.iq_scale + (1 << (k.iq_shift - 1))) >> k.iq_shift;
    else deq = (lvl * k.iq_scale) << -k.iq_shift;
    deq = (short)d_clip3(deq, -32768, 32767);
    const int err = a - deq;
    return ((long long)err * err) << k.cost_scale;
  };
  long long d_best = dist_of(q);
  if (q > 1) {
    const long long d1 = dist_of(q - 1);
    d_best = d1 < d_best ? d1 : d_best;
  }
  const long long zd = ((long long)(a * a)) << k.cost_scale;
  const long long coded = d_best + rq_bit_cost(sig1 + lvl_min, k.lambda);
  const long long coded_last = d_best + rq_bit_cost(lvl_min, k.lambda);
  gain = zd - (coded < zd ? coded : zd);
  gain_last = zd - coded_last;
  const unsigned lp = rq_last_pos_bits(cb, luma, w, h, k.scan_order, x, y);
  const int cbf_ctx = 2 * (!luma ? RQ_OFF(cbf_chroma)
                                 : (k.intra_cu ? RQ_OFF(cbf_luma) : RQ_OFF(root_cbf)));
  rhs = rq_bit_cost(cb[cbf_ctx + 1], k.lambda) - rq_bit_cost(cb[cbf_ctx], k.lambda) +
        rq_bit_cost(lp, k.lambda);
  // scan index: sub-block in the grid's scan, offset in the sub-block's (the
  // inverse of the 4x4 scan: position y * 4 + x -> scan offset, 16 nibbles)
  constexpr unsigned long long inv0 = rq_pack_scan4_inv(0), inv1 = rq_pack_scan4_inv(1),
                               inv2 = rq_pack_scan4_inv(2);
  const unsigned long long inv = k.scan_order == 0 ? inv0 : (k.scan_order == 1 ? inv1 : inv2);
  const int kk = (int)((inv >> (4 * (((y & 3) << 2) | (x & 3)))) & 15ull);
  idx = (d_sb_scan_index(k.scan_order, w >> 2, h >> 2, x >> 2, y >> 2) << 4) + kk;
}

// A snapshot's bit costs into cb[2 * 152] by the first `lanes` lanes (a word of
// four contexts per lane and round, its eight entries).
__device__ __forceinline__ void rq_prove_stage_costs(const xvcgpu_rdoq_contexts *snap,
                                                     unsigned *cb, int lane, int lanes) {
  constexpr int kWords = (int)sizeof(xvcgpu_rdoq_contexts) / 4;
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(snap);
  for (int wi = lane; wi < kWords; wi += lanes) {
    const uint32_t word = cw[wi];
    unsigned e[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned st8 = (word >> (8 * k)) & 127;
      e[2 * k] = gEntropyBits[st8];
      e[2 * k + 1] = gEntropyBits[st8 ^ 1];
    }
    uint4 *o = reinterpret_cast<uint4 *>(cb + 8 * wi);
    o[0] = make_uint4(e[0], e[1], e[2], e[3]);
    o[1] = make_uint4(e[4], e[5], e[6], e[7]);
  }
}

__global__ void __launch_bounds__(256)
rdoq_prove_zero_kernel(int bd, const xvcgpu_tx_block *blocks, int n, const int16_t *coeffs,
                       const uint32_t *d_off, int16_t *levels, int32_t *nnz_out,
                       const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm,
                       RdoqLists l) {
  __shared__ __attribute__((aligned(16))) unsigned s_cb[2 * sizeof(xvcgpu_rdoq_contexts)];
  __shared__ unsigned short s_xy[16][RQ_PROVE_MAX_CANDS];
  __shared__ int s_idx[16][RQ_PROVE_MAX_CANDS];
  __shared__ long long s_gain[16][RQ_PROVE_MAX_CANDS];
  __shared__ int s_n[16];
  __shared__ int s_fail[16];
  __shared__ int s_ctx;
  const int g = threadIdx.x >> 4, gl = threadIdx.x & 15;
  const int base = blockIdx.x * RQ_PROVE_BLOCKS;
  // the workgroup's snapshot: the largest index among its live blocks
  if (threadIdx.x == 0) s_ctx = -1;
  __syncthreads();
  if ((int)threadIdx.x < RQ_PROVE_BLOCKS) {
    const int b2 = base + (int)threadIdx.x;
    if (b2 < n && l.cls[b2] >= 0) atomicMax(&s_ctx, (int)rq_prm[b2].ctx_index);
  }
  __syncthreads();
  const int wg_ctx = s_ctx;
  if (wg_ctx < 0) return;   // no live block (uniform)
  rq_prove_stage_costs(&rq_ctx[wg_ctx], s_cb, (int)threadIdx.x, 256);
  __syncthreads();   // the context costs are in place
  // From here on a group of sixteen lanes works on its own block and its own rows
  // of the shared arrays: groups never share data, a wave holds four of them, so
  // ordering LDS traffic inside the wave is all that is needed.
  for (int sub = 0; sub < RQ_PROVE_BLOCKS / 16; sub++) {
    const int bi = base + sub * 16 + g;
    const bool live = bi < n && l.cls[bi < n ? bi : 0] >= 0;
    const xvcgpu_tx_block b = blocks[live ? bi : 0];
    const xvcgpu_rdoq_params prm = rq_prm[live ? bi : 0];
    if (gl == 0) {
      s_n[g] = 0;
      s_fail[g] = 0;
    }
    wave_sync();
    const RqProveBlock k = rq_prove_block(b, bd, prm);
    const int w = k.w, h = k.h;
    const int16_t *src = coeffs + d_off[live ? bi : 0];
    // (coefficients are read four at a time: a block that does not start on 8 bytes
    // is left to the walk)
    const bool tried = live && (int)prm.ctx_index == wg_ctx && w >= 4 && h >= 4 && w <= 32 &&
                       h <= 32 && (reinterpret_cast<uintptr_t>(src) & 7) == 0;
    auto qat = [&](int x, int y) {
      if (x >= w || y >= h) return 0;
      return rq_prove_quant(k, (short)d_abs((int)src[y * w + x]));
    };
    // the candidates: coefficients with q > 0 - one compare against the smallest
    // such magnitude, four coefficients per load (a magnitude of 32768: no proof)
    if (tried) {
      const int thr = rq_prove_threshold(k);
      bool wrap = false;
      for (int i = 4 * gl; i < w * h; i += 64) {
        const uint2 v = *reinterpret_cast<const uint2 *>(src + i);
        const int a4[4] = {d_abs((int)(short)(v.x & 0xffff)), d_abs((int)(short)(v.x >> 16)),
                           d_abs((int)(short)(v.y & 0xffff)), d_abs((int)(short)(v.y >> 16))};
#pragma unroll
        for (int t = 0; t < 4; t++) {
          wrap |= a4[t] == 32768;
          if (a4[t] >= thr) {
            const int slot = atomicAdd(&s_n[g], 1);
            const int ii = i + t;
            if (slot < RQ_PROVE_MAX_CANDS)
              s_xy[g][slot] = (unsigned short)(((ii >> k.lw) << 8) | (ii & (w - 1)));
          }
        }
      }
      if (wrap) s_fail[g] = 1;
    }
    wave_sync();   // the group's candidate list is complete
    const int nq = s_n[g];
    long long gain_last = 0, rhs = 0;
    int my_idx = 0;
    const bool cand = tried && nq <= RQ_PROVE_MAX_CANDS && gl < nq && !s_fail[g];
    if (tried && gl == 0 && (nq > RQ_PROVE_MAX_CANDS || nq == 0)) s_fail[g] = 1;
    if (cand) {
      const int x = s_xy[g][gl] & 255, y = s_xy[g][gl] >> 8;
      const int a = (short)d_abs((int)src[y * w + x]);
      long long gain;
      rq_prove_candidate(k, s_cb, x, y, a, qat, gain, gain_last, rhs, my_idx);
      s_idx[g][gl] = my_idx;
      s_gain[g][gl] = gain;
    }
    wave_sync();
    if (cand) {
      long long before = 0;
      for (int j = 0; j < nq; j++)
        if (s_idx[g][j] < my_idx) before += s_gain[g][j];
      if (before + gain_last >= rhs) s_fail[g] = 1;
    }
    wave_sync();
    if (tried && !s_fail[g]) {
      // QuantRdo would return 0: zero levels, no class
      int16_t *dst = levels + d_off[bi];
      for (int i = gl; i < w * h; i += 16) dst[i] = 0;
      if (gl == 0) {
        if (nnz_out) nnz_out[bi] = 0;
        l.cls[bi] = -1;
      }
    }
    wave_sync();   // the group's rows are free for its next block
  }
}

// The same proof where the coefficients of a block lie in LDS (the forward
// transform of the frame pass, k_tx2.h: C[x][y] at c[x * h + y]): G lanes per
// block, 64 / G blocks per wave; `cands` = the block's q > 0 coefficients already
// collected by the caller (count n_c, positions (y << 8) | x), wrap = a magnitude
// of 32768 was seen.  Returns true (to every lane of the group) when QuantRdo is
// bound to return 0.  The wave's LDS: the snapshot's costs, staged on first use.
struct RqProveLds {
  __attribute__((aligned(16))) unsigned cb[2 * sizeof(xvcgpu_rdoq_contexts)];
  long long gain[2][RQ_PROVE_MAX_CANDS];
  int idx[2][RQ_PROVE_MAX_CANDS];
  unsigned short xy[2][RQ_PROVE_MAX_CANDS];
  int n[2];
  int fail[2];
  int staged_ctx;   // snapshot whose costs cb holds (-1: none yet)
};

template <int G>
__device__ __forceinline__ bool rq_prove_zero_lds(RqProveLds &pv, const xvcgpu_tx_block &b, int bd,
                                                  const xvcgpu_rdoq_contexts *rq_ctx,
                                                  const xvcgpu_rdoq_params &prm,
                                                  const int16_t *c, bool live) {
  const int wl = (int)(threadIdx.x & 63), lane = wl & (G - 1), grp = wl / G;
  const int w = b.w, h = b.h;
  const int nq = pv.n[grp];
  bool tried = live && nq >= 1 && nq <= RQ_PROVE_MAX_CANDS && !pv.fail[grp] && w >= 4 && h >= 4 &&
               w <= 32 && h <= 32;
  // the wave's snapshot: the first trying group's
  const unsigned long long tm = __ballot(tried);
  if (!tm) return false;
  const int wave_ctx = __shfl((int)prm.ctx_index, __ffsll((long long)tm) - 1, 64);
  tried = tried && (int)prm.ctx_index == wave_ctx;
  if (pv.staged_ctx != wave_ctx) {
    wave_sync();
    rq_prove_stage_costs(&rq_ctx[wave_ctx], pv.cb, wl, 64);
    if (wl == 0) pv.staged_ctx = wave_ctx;
    wave_sync();
  }
  const RqProveBlock k = rq_prove_block(b, bd, prm);
  auto qat = [&](int x, int y) {
    if (x >= w || y >= h) return 0;
    return rq_prove_quant(k, (short)d_abs((int)c[x * h + y]));
  };
  long long gain_last = 0, rhs = 0;
  int my_idx = 0;
  const bool cand = tried && lane < nq;
  if (cand) {
    const int x = pv.xy[grp][lane] & 255, y = pv.xy[grp][lane] >> 8;
    const int a = (short)d_abs((int)c[x * h + y]);
    long long gain;
    rq_prove_candidate(k, pv.cb, x, y, a, qat, gain, gain_last, rhs, my_idx);
    pv.idx[grp][lane] = my_idx;
    pv.gain[grp][lane] = gain;
  }
  wave_sync();
  if (cand) {
    long long before = 0;
    for (int j = 0; j < nq; j++)
      if (pv.idx[grp][j] < my_idx) before += pv.gain[grp][j];
    if (before + gain_last >= rhs) pv.fail[grp] = 1;
  }
  wave_sync();
  return tried && !pv.fail[grp];
}

// The class lists from the per-block classes, without atomics (a few thousand
// atomicAdds on three addresses took 150 us): one workgroup, every thread
// counts its contiguous chunk, an LDS scan gives the chunk's place in each list.
// grid: 1; block: 1024.
__device__ __forceinline__ void rdoq_compact_kernel_body(int n, RdoqLists l) {
  __shared__ int part[3][1024];
  const int t = threadIdx.x;
  // a multiple of 4 blocks per thread: the classes are read four at a time
  const int per = ((n + 1023) / 1024 + 3) & ~3;
  const int a = t * per, e = a + per < n ? a + per : n;
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(l.cls);
  int cnt[3] = {0, 0, 0};
  for (int i = a; i < e; i += 4) {
    const uint32_t v = cw[i >> 2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)(signed char)(v >> (8 * k));
      const bool in = i + k < e;
      cnt[0] += in && c == 0;
      cnt[1] += in && c == 1;
      cnt[2] += in && c == 2;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) part[k][t] = cnt[k];
  __syncthreads();
  // inclusive scan over the 1024 partial counts, three at once: inside each
  // wave by shuffles, then the 16 wave totals by the first wave
  int inc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int v = cnt[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if ((t & 63) >= d) v += o;
    }
    inc[k] = v;
    if ((t & 63) == 63) part[k][t >> 6] = v;  // wave totals (slots 0..15 reused after the sync)
  }
  __syncthreads();
  if (t < 64) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      int v = t < 16 ? part[k][t] : 0;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (t >= d) v += o;
      }
      if (t < 16) part[k][16 + t] = v;  // inclusive totals of waves 0..t
    }
  }
  __syncthreads();
  int pos[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    pos[k] = inc[k] - cnt[k] + ((t >> 6) ? part[k][16 + (t >> 6) - 1] : 0);
  for (int i = a; i < e; i += 4) {
    const uint32_t v = cw[i >> 2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)(signed char)(v >> (8 * k));
      if (i + k < e && c >= 0) l.list[c][pos[c]++] = i + k;
    }
  }
  if (t == 1023)
    for (int k = 0; k < 3; k++) l.count[k] = part[k][16 + 15];
}

__global__ void __launch_bounds__(1024)
rdoq_compact_kernel(int n, RdoqLists l) {
  rdoq_compact_kernel_body(n, l);
}

// The same lists from many workgroups, still without atomics and in block order
// (the single workgroup above takes 13 us for a 1080p picture but 90 / 300 us for
// 2160p / 4320p - on every picture's critical path): a workgroup owns a chunk of
// RDOQ_CHUNK blocks, four per thread.  Launch 1 counts the chunk's classes; launch
// 2 sums the counts of the chunks in front of it (at most a few hundred), scans
// its own threads' counts and scatters.  grid: ceil(n / RDOQ_CHUNK); block: 1024.
__device__ __forceinline__ void rdoq_chunk_counts(int n, const RdoqLists &l, int cnt[3]) {
  const int i = (int)blockIdx.x * RDOQ_CHUNK + 4 * (int)threadIdx.x;
  cnt[0] = cnt[1] = cnt[2] = 0;
  if (i < n) {
    const uint32_t v = reinterpret_cast<const uint32_t *>(l.cls)[i >> 2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)(signed char)(v >> (8 * k));
      const bool in = i + k < n;
      cnt[0] += in && c == 0;
      cnt[1] += in && c == 1;
      cnt[2] += in && c == 2;
    }
  }
}

__global__ void __launch_bounds__(1024)
rdoq_count_kernel(int n, RdoqLists l) {
  __shared__ int wsum[3][16];
  int cnt[3];
  rdoq_chunk_counts(n, l, cnt);
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int v = wave_reduce_add_i32(cnt[k]);
    if ((t & 63) == 0) wsum[k][t >> 6] = v;
  }
  __syncthreads();
  if (t < 3) {
    int v = 0;
    for (int w = 0; w < 16; w++) v += wsum[t][w];
    l.part[4 * blockIdx.x + t] = v;
  }
}

__global__ void __launch_bounds__(1024)
rdoq_scatter_kernel(int n, RdoqLists l) {
  __shared__ int part[3][1024];
  __shared__ int base[3];
  const int t = threadIdx.x, b = blockIdx.x;
  // the chunks in front of this one
  {
    int v[3] = {0, 0, 0};
    for (int j = t; j < b; j += 1024) {
      v[0] += l.part[4 * j];
      v[1] += l.part[4 * j + 1];
      v[2] += l.part[4 * j + 2];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int w = wave_reduce_add_i32(v[k]);
      if ((t & 63) == 0) part[k][t >> 6] = w;
    }
    __syncthreads();
    if (t < 3) {
      int sum = 0;
      for (int w = 0; w < 16; w++) sum += part[t][w];
      base[t] = sum;
    }
    __syncthreads();
  }
  int cnt[3];
  rdoq_chunk_counts(n, l, cnt);
  int inc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int v = cnt[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if ((t & 63) >= d) v += o;
    }
    inc[k] = v;
  }
  __syncthreads();   // base[] read by everybody before part[] is reused
  const int b0 = base[0], b1 = base[1], b2 = base[2];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; k++)
    if ((t & 63) == 63) part[k][t >> 6] = inc[k];
  __syncthreads();
  if (t < 64) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      int v = t < 16 ? part[k][t] : 0;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (t >= d) v += o;
      }
      if (t < 16) part[k][16 + t] = v;
    }
  }
  __syncthreads();
  const int bs[3] = {b0, b1, b2};
  int pos[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    pos[k] = bs[k] + inc[k] - cnt[k] + ((t >> 6) ? part[k][16 + (t >> 6) - 1] : 0);
  const int i = b * RDOQ_CHUNK + 4 * t;
  if (i < n) {
    const uint32_t v = reinterpret_cast<const uint32_t *>(l.cls)[i >> 2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = (int)(signed char)(v >> (8 * k));
      if (i + k < n && c >= 0) l.list[c][pos[c]++] = i + k;
    }
  }
  if (b == (int)gridDim.x - 1 && t == 1023)
    for (int k = 0; k < 3; k++) l.count[k] = bs[k] + part[k][16 + 15];
}

// LDS of one wave: G lanes per block (64 / G blocks per wave), blocks of at most NSB
// region sub-blocks.  One table of context costs per wave: the groups of a wave
// nearly always name the same snapshot; when they do not, the wave serves one
// snapshot after the other (a table per group cost 19 KB of LDS per wave in the
// 4-lane class - enough, with the walk's long-lived waves, to keep other kernels
// off the CUs).
template <int G, int NSB>
struct RdoqPackedLds {
  static constexpr int GROUPS = 64 / G;
  static constexpr int MAXC = 16 * NSB;                                // region coefficients
  static constexpr int MAXSB = NSB;                                    // region sub-blocks
  static constexpr int MAXR = RQ_PADDED(MAXC);     // records: RQ_SB_STRIDE per sub-block
  static constexpr int MAXT = RQ_CF_PADDED(MAXC);  // coefficient / level tiles: RQ_CF_STRIDE
  // the per-coefficient records: 5 bytes each (+ 4 of coefficient and level)
  alignas(8) long long sb_code_cost[GROUPS][MAXSB];
  alignas(8) int16_t cf[GROUPS][MAXT], lv[GROUPS][MAXT];
  alignas(8) int16_t wl[GROUPS][RQ_WL(MAXC)];          // wave_rdoq4's working levels
  long long fcs[NSB == 64 ? 256 : 1];                  // (64-point sides; else sb_code_cost)
  unsigned short rate_up[GROUPS][MAXR];
  alignas(8) unsigned ctx_bits[2 * sizeof(xvcgpu_rdoq_contexts)];
  unsigned csbf_bits[GROUPS][MAXSB];
  unsigned char csbf[GROUPS][MAXSB], sb_live[GROUPS][MAXSB], sb_dcz[GROUPS][MAXSB];
  // the whole grid's scan (64-point sides: up to four times the region; NSB = 4: grid = region)
  unsigned char sb_of_scan[GROUPS][NSB == 4 ? MAXSB : MAXSB * 4];
  // one table per wave: the groups of a round share shape, scan and snapshot
  unsigned lp_bits[32];
};

// G lanes per block; grid: an upper bound on ceil(count / (64 / G)) waves (the
// list's count is read on the device); block: 64.
// wave_rdoq4's accessors in the packed kernel: the coefficient tile (sub-block major,
// RQ_CF_STRIDE: a sub-block's row is 8 contiguous bytes) and the block's levels in
// global memory (row-major, the block's own width): a unit's row is one access.
struct RqTileCf {
  const int16_t *p;
  int rgw;
  __device__ __forceinline__ int pos(int x, int y) const {
    return ((y >> 2) * rgw + (x >> 2)) * RQ_CF_STRIDE + (((y & 3) << 2) | (x & 3));
  }
  __device__ __forceinline__ int operator()(int x, int y) const { return (int)p[pos(x, y)]; }
  __device__ __forceinline__ void row4(int px, int y, int c[4]) const {
    const uint2 v = *reinterpret_cast<const uint2 *>(p + pos(px, y));
    c[0] = (int)(short)(v.x & 0xffffu);
    c[1] = (int)(short)(v.x >> 16);
    c[2] = (int)(short)(v.y & 0xffffu);
    c[3] = (int)(short)(v.y >> 16);
  }
};
struct RqGlobalLev {
  int16_t *p;
  int w;
  __device__ __forceinline__ int16_t *operator()(int x, int y) const { return p + y * w + x; }
  __device__ __forceinline__ void store4(int px, int y, const int v[4]) const {
    int16_t *o = p + y * w + px;
    if ((reinterpret_cast<uintptr_t>(o) & 7) == 0) {
      *reinterpret_cast<uint2 *>(o) =
          make_uint2((unsigned)(v[0] & 0xffff) | ((unsigned)v[1] << 16),
                     (unsigned)(v[2] & 0xffff) | ((unsigned)v[3] << 16));
    } else {
      o[0] = (int16_t)v[0]; o[1] = (int16_t)v[1]; o[2] = (int16_t)v[2]; o[3] = (int16_t)v[3];
    }
  }
};

template <int G, int NSB, bool FOUR>
__device__ __forceinline__ void quant_rdo_packed_wave(
    RdoqPackedLds<G, NSB> &sm, int wave, int bd, const xvcgpu_tx_block *blocks, const int *list,
    const int *count, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels,
    int32_t *nnz_out, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm,
    xvcgpu_cu_info *cu_patch = nullptr) {
  constexpr int GROUPS = 64 / G;
  const int g = threadIdx.x / G, lane = threadIdx.x % G;
  const int slot = wave * GROUPS + g;
  RQ_TRACE(0);
  RQ_TRACE_RT(0);
  const int n_list = *count;
  if (wave * GROUPS >= n_list) return;  // the launch is an upper bound
  const bool active = slot < n_list;
  const int bi = active ? list[slot] : 0;
  const xvcgpu_tx_block b = blocks[bi];
  const xvcgpu_rdoq_params prm = rq_prm[bi];
  const int w = b.w, h = b.h;
  const int rw = w < 32 ? w : 32, rh = h < 32 ? h : 32;
  const uint32_t off = d_off[bi];
  const int16_t *src = coeffs + off;
  int16_t *cf = sm.cf[g], *lv = sm.lv[g];
  // this lane's word of its own block's snapshot, fetched beside the coefficients (the
  // round below nearly always serves that snapshot: one global round trip less)
  constexpr int kCtxWords = (int)sizeof(xvcgpu_rdoq_contexts) / 4;
  const int ctx_wi = (int)threadIdx.x < kCtxWords ? (int)threadIdx.x : kCtxWords - 1;
  const uint32_t word_own = reinterpret_cast<const uint32_t *>(&rq_ctx[prm.ctx_index])[ctx_wi];
  RQ_TRACE(1);
  // blocks that are their own region (no 64-point side), a multiple of 8
  // coefficients, 16-byte aligned in both arrays: 16-byte copies in and out
  int16_t *dst = levels + off;
  const bool wide = w == rw && h == rh && ((w * h) & 7) == 0 &&
                    ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  // the tiles are stored sub-block by sub-block (stride RQ_CF_STRIDE: the lanes
  // of a group read the same offset of different sub-blocks - RQ_SB_STRIDE above);
  // blocks with 2x2 sub-blocks (a side of 2) stay raster
  const bool sb4 = !(w == 2 || h == 2);
  const int rgw4 = rw >> 2;
  auto tile_pos = [sb4, rgw4, rw](int x, int y) {
    return sb4 ? ((y >> 2) * rgw4 + (x >> 2)) * RQ_CF_STRIDE + (((y & 3) << 2) | (x & 3))
               : y * rw + x;
  };
  if (active) {
    if (wide && sb4) {
      // 8 coefficients = two sub-blocks' rows of 4 (a 4-wide block: two rows of one)
      for (int i = lane; i < (w * h) >> 3; i += G) {
        const uint4 v8 = reinterpret_cast<const uint4 *>(src)[i];
        const int y = (8 * i) / rw, x = 8 * i - y * rw;
        const int p0 = tile_pos(x, y), p1 = rw == 4 ? tile_pos(0, y + 1) : tile_pos(x + 4, y);
        *reinterpret_cast<uint2 *>(cf + p0) = make_uint2(v8.x, v8.y);
        *reinterpret_cast<uint2 *>(cf + p1) = make_uint2(v8.z, v8.w);
        if (!FOUR) {   // (wave_rdoq4 writes its levels to global memory itself)
          *reinterpret_cast<uint2 *>(lv + p0) = make_uint2(0, 0);
          *reinterpret_cast<uint2 *>(lv + p1) = make_uint2(0, 0);
        }
      }
    } else {
      for (int i = lane; i < rw * rh; i += G) {
        const int y = i / rw, x = i - y * rw;
        cf[tile_pos(x, y)] = src[y * w + x];
        if (!FOUR) lv[tile_pos(x, y)] = 0;
      }
    }
  }
  RdoqView v;
  v.wl = sm.wl[g];
  v.fcs = NSB == 64 ? sm.fcs : sm.sb_code_cost[g];
  v.sb_dcz = sm.sb_dcz[g];
  v.rate_up = sm.rate_up[g];
  v.sb_live = sm.sb_live[g];
  v.sb_code_cost = sm.sb_code_cost[g];
  v.csbf_bits = sm.csbf_bits[g];
  v.csbf = sm.csbf[g];
  v.sb_of_scan = sm.sb_of_scan[g];
  v.lp_bits = sm.lp_bits;
  v.ctx_bits = sm.ctx_bits;
  RQ_TRACE(2);
  // One (context snapshot, block shape, qp, component kind, flags, lambda) at a
  // time - normally one round: a picture's blocks of one class share them.  What
  // the walk derives from them (quantiser scales and shifts, grid sizes, context
  // bases, the lambda it multiplies every rate with) then lives in SCALAR
  // registers; as per-lane values - every group of a wave its own block - they
  // cost the walk 70 of its 200 vector registers, i.e. a wave per SIMD.
  bool pending = active;
  int nnz = 0;
  for (;;) {
    const unsigned long long todo = __ballot(pending);
    if (!todo) break;
    const int leader = __ffsll((long long)todo) - 1;
    const int cur = __shfl((int)prm.ctx_index, leader, 64);
    // the leader's key, as per-lane copies for the comparison
    const int kw = __shfl(w, leader, 64), kh = __shfl(h, leader, 64);
    const int kqp = __shfl((int)b.qp, leader, 64);
    const int kluma = __shfl((int)(b.comp == 0), leader, 64);
    const int kflags = __shfl((int)b.intra_pic, leader, 64);
    const int kpf = __shfl((int)prm.flags, leader, 64);
    const long long klambda = __shfl(prm.lambda, leader, 64);
    wave_sync();  // the previous round's readers are done with the table
    {
      // two round trips in all: a word of four contexts per lane, then its eight
      // table entries (byte by byte it was three dependent pairs of trips)
      static_assert(sizeof(xvcgpu_rdoq_contexts) % 4 == 0 &&
                        sizeof(xvcgpu_rdoq_contexts) / 4 <= 64,
                    "one word of contexts per lane");
      constexpr int kWords = (int)sizeof(xvcgpu_rdoq_contexts) / 4;
      const uint32_t *cw = reinterpret_cast<const uint32_t *>(&rq_ctx[cur]);
      const uint32_t word = (int)prm.ctx_index == cur ? word_own : cw[ctx_wi];
      unsigned e[8];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned st8 = (word >> (8 * k)) & 127;
        e[2 * k] = kEntropyBits[st8];
        e[2 * k + 1] = kEntropyBits[st8 ^ 1];
      }
      if ((int)threadIdx.x < kWords) {
        uint4 *o = reinterpret_cast<uint4 *>(sm.ctx_bits + 8 * threadIdx.x);
        o[0] = make_uint4(e[0], e[1], e[2], e[3]);
        o[1] = make_uint4(e[4], e[5], e[6], e[7]);
      }
    }
    wave_sync();
    RQ_TRACE(3);
    const bool take = pending && (int)prm.ctx_index == cur && w == kw && h == kh &&
                      (int)b.qp == kqp && (int)(b.comp == 0) == kluma &&
                      (int)b.intra_pic == kflags && (int)prm.flags == kpf &&
                      prm.lambda == klambda;
    if (take) {
      // every lane in here holds the same key: read it into scalar registers
      const int uw = __builtin_amdgcn_readfirstlane(w), uh = __builtin_amdgcn_readfirstlane(h);
      const int uqp = __builtin_amdgcn_readfirstlane((int)b.qp);
      const int uluma = __builtin_amdgcn_readfirstlane((int)(b.comp == 0));
      const int uflags = __builtin_amdgcn_readfirstlane((int)b.intra_pic);
      const int upf = __builtin_amdgcn_readfirstlane((int)prm.flags);
      const unsigned ulam_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)prm.lambda);
      const unsigned ulam_hi = (unsigned)__builtin_amdgcn_readfirstlane(
          (int)(unsigned)((unsigned long long)prm.lambda >> 32));
      const long long ulambda = (long long)(((unsigned long long)ulam_hi << 32) | ulam_lo);
      // the groups walk independently (their shuffles stay inside the group)
      const bool usb4 = !(uw == 2 || uh == 2);
      const int urw = uw < 32 ? uw : 32, urgw4 = urw >> 2;
      auto utile = [usb4, urgw4, urw](int x, int y) {
        return usb4 ? ((y >> 2) * urgw4 + (x >> 2)) * RQ_CF_STRIDE + (((y & 3) << 2) | (x & 3))
                    : y * urw + x;
      };
      xvcgpu_rdoq_params uprm = prm;   // rd_factor stays the lane's own
      uprm.lambda = ulambda;
      uprm.flags = (uint8_t)upf;
      const int uscan = (uflags >> XVC_TXF_SCAN_SHIFT) & 3;
      auto cf_at = [cf, utile](int x, int y) { return (int)cf[utile(x, y)]; };
      auto lv_at = [lv, utile](int x, int y) { return lv + utile(x, y); };
      const bool ush = !(uflags & XVC_TXF_NO_SIGN_HIDING);
      if (FOUR) {
        // four lanes per sub-block (k_rdoq4.h): the class lists hold what it takes; the
        // levels go straight to the block's place in global memory
        const RqTileCf cf4 = {cf, urgw4};
        const RqGlobalLev lv4 = {dst, uw};
        nnz = wave_rdoq4<G, 1>(v, lane, bd, uw, uh, uqp, uluma != 0, ush, uprm, cf4, lv4);
      } else if (NSB == 64 && rq4_takes(64, uw, uh, uscan)) {
        // (more than sixteen sub-blocks or a 64-point side: four units per lane)
        nnz = wave_rdoq4<64, 4>(v, lane, bd, uw, uh, uqp, uluma != 0, ush, uprm, cf_at, lv_at);
      } else if (G == 16 && NSB == 16 && uw == 16 && uh == 16) {
        // (the instance for the exact 16x16 block beside the any-size one: the long lists of
        // a large picture - RDOQ4_LATENCY_BLOCKS - are almost only these)
        auto cf16 = [cf](int x, int y) {
          return (int)cf[((y >> 2) * 4 + (x >> 2)) * RQ_CF_STRIDE + (((y & 3) << 2) | (x & 3))];
        };
        auto lv16 = [lv](int x, int y) {
          return lv + ((y >> 2) * 4 + (x >> 2)) * RQ_CF_STRIDE + (((y & 3) << 2) | (x & 3));
        };
        nnz = wave_rdoq<G>(v, lane, bd, 16, 16, uqp, uluma != 0, uscan, ush, rq_ctx[cur], uprm,
                           cf16, lv16, false, false);
      } else {
        nnz = wave_rdoq<G>(v, lane, bd, uw, uh, uqp, uluma != 0, uscan, ush, rq_ctx[cur], uprm,
                           cf_at, lv_at, false, false);
      }
      pending = false;
    }
  }
  wave_sync();
  RQ_TRACE(9);
  if (!active) return;
  if (FOUR) {
    // (the walk wrote the levels)
  } else if (wide && sb4) {
    for (int i = lane; i < (w * h) >> 3; i += G) {
      const int y = (8 * i) / rw, x = 8 * i - y * rw;
      const uint2 a = *reinterpret_cast<const uint2 *>(lv + tile_pos(x, y));
      const uint2 b2 = *reinterpret_cast<const uint2 *>(
          lv + (rw == 4 ? tile_pos(0, y + 1) : tile_pos(x + 4, y)));
      reinterpret_cast<uint4 *>(dst)[i] = make_uint4(a.x, a.y, b2.x, b2.y);
    }
  } else {
    for (int i = lane; i < w * h; i += G) {
      const int y = i / w, x = i - y * w;
      dst[i] = (x < rw && y < rh) ? lv[tile_pos(x, y)] : (int16_t)0;
    }
  }
  if (lane == 0 && nnz_out) nnz_out[bi] = nnz;
  // blocks 3 * cu + comp (xvcgpu_fwd_from_me_classify wrote the CU's record with cbf_luma = 0)
  if (lane == 0 && cu_patch && b.comp == 0 && nnz) cu_patch[bi / 3].cbf_luma = 1;
  RQ_TRACE(10);
  RQ_TRACE_RT(1);
}

// The classes' walks are independent and each is bounded by its own slowest block.
// Two launches: the two classes of wave_rdoq4 together (quant_rdo_packed4_kernel:
// workgroups [0, g16) take the blocks of up to sixteen sub-blocks, a block per wave,
// the rest the blocks of up to four, four per wave), and the general class
// (quant_rdo_packed_kernel, a lane per sub-block) - in one kernel the general walk's
// 198 vector registers were every wave's: two waves per SIMD, 2048 wave slots on the
// chip for the 2300 - 7600 blocks of a 1080p picture, i.e. a second round of waves
// behind the first (66 us for walks of at most 45).  The lists' counts are only known
// on the device, and one workgroup per possible block made the launch last as long as
// the churn of workgroups retiring after a look at the count: each class gets a bounded
// number of workgroups instead, which walk their list with that stride.
// grid: g16 + g4 (packed4) / g64 (general); block: 64.
#ifndef RDOQ4_LATENCY_BLOCKS
#define RDOQ4_LATENCY_BLOCKS 4096   // a block per wave up to this many blocks of the class
#endif
__device__ __forceinline__ void quant_rdo_packed4_kernel_body(int bd, const xvcgpu_tx_block *blocks, RdoqLists l, int g16, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, xvcgpu_cu_info *cu_patch) {
  union Lds {
    RdoqPackedLds<64, 16> a;
    RdoqPackedLds<16, 4> b;
    RdoqPackedLds<16, 16> t;
  };
  __shared__ Lds sm;
  const int wg = blockIdx.x, g4 = (int)gridDim.x - g16;
  if (wg < g16) {
    const int n1 = l.count[1];
    if (n1 > RDOQ4_LATENCY_BLOCKS) {
      // More blocks than the chip has wave slots: waves queue behind each other whatever
      // their shape, and what the launch costs - alone and beside the other pictures'
      // kernels - is its total instruction count.  A block per wave spends about twice
      // the wave instructions per block of the lane-per-sub-block walk with four blocks
      // side by side (2160p QP 27, 25 000 blocks, three chains: 1290 against 1550 frame
      // passes/s): the long lists take that walk.
      const int waves = (n1 + 3) >> 2;
      for (int wv = wg; wv < waves; wv += g16) {
        quant_rdo_packed_wave<16, 16, false>(sm.t, wv, bd, blocks, l.list[1], l.count + 1, coeffs,
                                             d_off, levels, nnz_out, rq_ctx, rq_prm, cu_patch);
        wave_sync();
      }
      return;
    }
    for (int wv = wg; wv < n1; wv += g16) {
      quant_rdo_packed_wave<64, 16, true>(sm.a, wv, bd, blocks, l.list[1], l.count + 1, coeffs,
                                          d_off, levels, nnz_out, rq_ctx, rq_prm, cu_patch);
      wave_sync();
    }
  } else {
    const int waves = (l.count[0] + 3) >> 2;
    for (int wv = wg - g16; wv < waves; wv += g4) {
      quant_rdo_packed_wave<16, 4, true>(sm.b, wv, bd, blocks, l.list[0], l.count + 0, coeffs,
                                         d_off, levels, nnz_out, rq_ctx, rq_prm, cu_patch);
      wave_sync();
    }
  }
}

__device__ __forceinline__ void quant_rdo_packed_kernel_body(int bd, const xvcgpu_tx_block *blocks, RdoqLists l, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, xvcgpu_cu_info *cu_patch) {
  __shared__ RdoqPackedLds<64, 64> sm;
  const int waves = l.count[2];
  for (int wv = blockIdx.x; wv < waves; wv += (int)gridDim.x) {
    quant_rdo_packed_wave<64, 64, false>(sm, wv, bd, blocks, l.list[2], l.count + 2, coeffs, d_off,
                                         levels, nnz_out, rq_ctx, rq_prm, cu_patch);
    wave_sync();
  }
}

#ifndef RDOQ4_MIN_WAVES
#define RDOQ4_MIN_WAVES 3   // per SIMD: 168 vector registers (the walk holds 156 - 167)
#endif
__global__ void __launch_bounds__(64, RDOQ4_MIN_WAVES)
quant_rdo_packed4_kernel(int bd, const xvcgpu_tx_block *blocks, RdoqLists l, int g16, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, xvcgpu_cu_info *cu_patch = nullptr, int *misuse = nullptr) {
  // misuse: the general class's launch was left out on the caller's word
  // (xvcgpu_quant_rdo_set_four_lane_only); a block on its list is reported, not dropped
  if (misuse && blockIdx.x == 0 && threadIdx.x == 0 && l.count[2] != 0) *misuse = 1;
  quant_rdo_packed4_kernel_body(bd, blocks, l, g16, coeffs, d_off, levels, nnz_out, rq_ctx, rq_prm, cu_patch);
}

#ifndef RDOQ_MIN_WAVES
#define RDOQ_MIN_WAVES 2   // per SIMD: 256 vector registers
#endif
__global__ void __launch_bounds__(64, RDOQ_MIN_WAVES)
quant_rdo_packed_kernel(int bd, const xvcgpu_tx_block *blocks, RdoqLists l, const int16_t *coeffs, const uint32_t *d_off, int16_t *levels, int32_t *nnz_out, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, xvcgpu_cu_info *cu_patch = nullptr) {
  quant_rdo_packed_kernel_body(bd, blocks, l, coeffs, d_off, levels, nnz_out, rq_ctx, rq_prm, cu_patch);
}

#endif  // XVCGPU_K_RDOQ_H_
