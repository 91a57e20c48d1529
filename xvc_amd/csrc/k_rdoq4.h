// k_rdoq4.h -- RdoQuant::QuantRdo with FOUR lanes per 4x4 sub-block
// (xvc_enc_lib/rdo_quant.cc:224-446, QuantCoeffRdo :708-735, EvalZeroSubblock
// :737-775, EvalLastPos :777-850, CoeffSignHideRdo :575-705), for every block whose
// coefficients are walked in the diagonal scan with 4x4 sub-blocks: 4x4 .. 64x64
// (coefficients exist in the 32x32 low-frequency corner at most, transform.cc:1458).
// Same arithmetic and the same per-coefficient records as wave_rdoq (k_rdoq.h: one
// lane per sub-block), which keeps the horizontal / vertical scans and the 2-wide
// blocks.
//
// Why.  The launch of the packed quantiser lasts its slowest wave's life, and a lone
// wave issues an instruction every 8 - 10 clocks whatever its lanes do: the life is
// the wave's DYNAMIC INSTRUCTION COUNT.  With a lane per sub-block every loop over a
// sub-block's sixteen coefficients (quantise + last position, zero-out and signs,
// the sign hiding's two passes) is sixteen trips, the coefficients without a choice
// of one anti-diagonal of sub-blocks take four rounds of sixteen lanes, and four
// blocks share a wave, so every data-dependent loop runs as often as the worst of
// four blocks needs.  Here a "unit" = (sub-block, row of the sub-block), four
// consecutive lanes (a quad) per sub-block:
//   * a lane holds its row's four coefficients, plain quantised values and signs in
//     registers: the per-coefficient loops are four trips;
//   * the decisions of a sub-block run along ITS anti-diagonals: the template of a
//     coefficient (cabac.cc:535-552) only reads positions on the next two
//     anti-diagonals, and the coefficients of one anti-diagonal lie in four
//     different rows, i.e. on four lanes.  What the scan order still constrains
//     inside an anti-diagonal is the greater-1 / greater-2 budget (c1_idx < 8,
//     c2_idx < 1: all GetAbsLevelBits reads of it): a lane decides with the budget
//     as it stands in front of the anti-diagonal, the quad exchanges the levels,
//     and a lane whose budget CLASS differs once the levels in front of it are
//     counted decides again (the class changes at most twice per sub-block, so
//     this second trip is rare; lanes are final in scan order, four trips at most);
//   * one block per wave (G = 64): the no-choice coefficients of an anti-diagonal of
//     sub-blocks (4 x 16, or 8 x 16 in a 32x32 corner) are one or two rounds, and no
//     other block's trip counts are paid for;
//   * blocks of more than sixteen sub-blocks (NR = 4 units per lane: sub-block
//     16 r + lane / 4 for r = 0..3): the sub-blocks of ONE anti-diagonal of the
//     sub-block grid fall on different quads (their indices differ by multiples of
//     gw - 1, odd, never a multiple of 16 inside a diagonal), so a lane has at most
//     one unit to work on per diagonal;
//   * the levels being decided live in a raster array with a border of two zero
//     columns / rows to the right / below: a template is five reads at constant
//     offsets from the coefficient's address, no clamps, no masks; the final levels
//     are written to the caller's array once, signs re-applied;
//   * a coefficient's position class (x + y) is 4 d + the sub-block's anti-diagonal,
//     the same for every lane of a step: the context bases are scalar.
// 64-point sides: the sub-blocks beyond the corner never hold a level; they take part
// in the scan's numbering and cost a zero coded-sub-block flag each.
#ifndef XVCGPU_K_RDOQ4_H_
#define XVCGPU_K_RDOQ4_H_

// ---- quad (four consecutive lanes) exchanges: DPP quad_perm, no LDS ------------
template <int I>
__device__ __forceinline__ int rq4_quad_bcast(int v) {   // lane I of the quad to all four
  return __builtin_amdgcn_update_dpp(0, v, I * 0x55, 0xF, 0xF, false);
}
__device__ __forceinline__ int rq4_quad_or(int v) {
  v |= lane_xor<1>(v);
  v |= lane_xor<2>(v);
  return v;
}
__device__ __forceinline__ int rq4_quad_min(int v) {
  int o = lane_xor<1>(v);
  v = o < v ? o : v;
  o = lane_xor<2>(v);
  return o < v ? o : v;
}
__device__ __forceinline__ int rq4_quad_max(int v) {
  int o = lane_xor<1>(v);
  v = o > v ? o : v;
  o = lane_xor<2>(v);
  return o > v ? o : v;
}
// Reductions over the G (16 or 64) lanes of a block without the LDS crossbar
// (__shfl_xor is ds_bpermute: six dependent LDS round trips per 64-lane reduction, and
// the walk had fifteen of them): quad_perm, row_half_mirror and row_mirror inside a
// row of sixteen, the four rows through v_readlane - the result is wave-uniform
// (G = 64) or equal on the row's sixteen lanes (G = 16).
__device__ __forceinline__ int rq4_row_mirror(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
}
__device__ __forceinline__ int rq4_row_half_mirror(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
}
template <int G>
__device__ __forceinline__ int rq4_max(int v) {
  int o = lane_xor<1>(v);
  v = o > v ? o : v;
  o = lane_xor<2>(v);
  v = o > v ? o : v;
  o = rq4_row_half_mirror(v);
  v = o > v ? o : v;
  o = rq4_row_mirror(v);
  v = o > v ? o : v;
  if (G == 64) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
              c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    v = ab > cd ? ab : cd;
  }
  return v;
}
template <int G>
__device__ __forceinline__ int rq4_sum(int v) {
  v = dpp_group_sum<16>(v);
  if (G == 64)
    v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
        __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
  return v;
}
__device__ __forceinline__ long long rq4_readlane_i64(long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long rq4_add_wrap(long long a, long long b) {
  return (long long)((unsigned long long)a + (unsigned long long)b);
}
template <int G>
__device__ __forceinline__ long long rq4_sum_i64(long long v) {
  v = rq_group_sum_i64<16>(v);
  if (G == 64)
    v = rq4_add_wrap(rq4_add_wrap(rq4_readlane_i64(v, 0), rq4_readlane_i64(v, 16)),
                     rq4_add_wrap(rq4_readlane_i64(v, 32), rq4_readlane_i64(v, 48)));
  return v;
}
// inclusive prefix sum over each row of sixteen lanes (lanes ascending), row_shr
__device__ __forceinline__ unsigned rq4_row_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  return v;
}
template <int S>
__device__ __forceinline__ long long rq4_xor_i64(long long v) {
  const int lo = lane_xor<S>((int)(unsigned)v), hi = lane_xor<S>((int)(v >> 32));
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Inclusive SUFFIX sum of v over the G lanes of a block (lanes ascending): the sum of
// the lanes at or behind this one.
template <int G>
__device__ __forceinline__ long long rq4_suffix_i64(long long v) {
  const long long inc = rq_row_scan_i64(v);            // prefix inside the row
  const long long row = rq_group_sum_i64<16>(v);       // the row's total
  long long behind = rq4_add_wrap(row - inc, v);
  if (G == 64) {
    const long long t1 = rq4_readlane_i64(row, 16), t2 = rq4_readlane_i64(row, 32),
                    t3 = rq4_readlane_i64(row, 48);
    const int r = ME2_LANE >> 4;
    behind = rq4_add_wrap(behind, r == 0 ? rq4_add_wrap(rq4_add_wrap(t1, t2), t3)
                                         : (r == 1 ? rq4_add_wrap(t2, t3) : (r == 2 ? t3 : 0ll)));
  }
  return behind;
}

// TransformHelper::DeriveSubblockScan's index of sub-block (sx, sy) in the diagonal
// scan over a gw x gh grid (d_sb_scan_index, k_tx.h) in closed form: the cells on the
// diagonals in front of s = sx + sy (their lengths grow to min(gw, gh), stay, shrink),
// then the place on its own diagonal.
__device__ __forceinline__ int rq4_sb_scan_index(int gw, int gh, int sx, int sy) {
  const int s = sx + sy, a = gw < gh ? gw : gh, b = gw < gh ? gh : gw;
  int t;
  if (s <= a) {
    t = (s * (s + 1)) >> 1;
  } else if (s <= b) {
    t = ((a * (a + 1)) >> 1) + (s - a) * a;
  } else {
    const int m = gw + gh - 1 - s;
    t = gw * gh - ((m * (m + 1)) >> 1);
  }
  return t + ((s < gh - 1 ? s : gh - 1) - sy);
}

// Blocks wave_rdoq4 takes (lanes per block G = 64: any such block; G = 16: the ones of
// at most four sub-blocks)
__device__ __forceinline__ bool rq4_takes(int G, int w, int h, int scan_order) {
  const int rw = w < 32 ? w : 32, rh = h < 32 ? h : 32;
  return scan_order == 0 && w >= 4 && h >= 4 && (G == 64 || (rw >> 2) * (rh >> 2) * 4 <= G);
}
// units per lane a block needs with 64 lanes: 1 (up to sixteen sub-blocks) or 4
__device__ __forceinline__ bool rq4_needs_nr4(int w, int h) {
  const int rw = w < 32 ? w : 32, rh = h < 32 ? h : 32;
  return (rw >> 2) * (rh >> 2) > 16;
}

// A row of four coefficients / levels at (px .. px + 3, y).  The generic forms go
// through the accessors' operator(); an accessor whose rows are contiguous can offer
// row4() / store4() (one 8-byte access).
template <typename CF>
__device__ __forceinline__ auto rq4_load_row(const CF &cf, int px, int y, int c[4])
    -> decltype(cf.row4(px, y, c), void()) {
  cf.row4(px, y, c);
}
template <typename CF, typename... Dummy>
__device__ __forceinline__ void rq4_load_row(const CF &cf, int px, int y, int c[4], Dummy...) {
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = cf(px + i, y);
}
template <typename LEV>
__device__ __forceinline__ auto rq4_store_row(const LEV &lev, int px, int y, const int v[4])
    -> decltype(lev.store4(px, y, v), void()) {
  lev.store4(px, y, v);
}
template <typename LEV, typename... Dummy>
__device__ __forceinline__ void rq4_store_row(const LEV &lev, int px, int y, const int v[4],
                                              Dummy...) {
#pragma unroll
  for (int i = 0; i < 4; i++) *lev(px + i, y) = (short)v[i];
}

// S: RdoqShared<N> or RdoqView (k_rdoq.h).  `lane` = 0..G-1, all G lanes call
// (G = 64: the whole wave; G = 16: four blocks of a wave side by side, w, h and
// the other scalar arguments then equal for the four, NR = 1).  s.ctx_bits holds the
// snapshot's bit costs (the caller staged them).  cf(x, y) reads a coefficient,
// lev(x, y) addresses the output level (both inside the corner only).  Returns the
// number of non-zero levels to every lane; every level of the corner is written
// (levels of a 64-point side beyond it are the caller's to clear).
template <int G, int NR, typename S, typename CF, typename LEV>
__device__ __forceinline__ int wave_rdoq4(S &s, int lane, int bd, int w, int h, int comp_qp,
                                          bool luma, bool sign_hide,
                                          const xvcgpu_rdoq_params &prm, CF cf, LEV lev) {
  static_assert(NR == 1 || (NR == 4 && G == 64), "one unit per lane, or four with 64 lanes");
  constexpr unsigned long long kInv = rq_pack_scan4_inv(0);   // y * 4 + x -> scan offset
  constexpr int UPR = G / 4;                                  // sub-blocks per round of units
  const int gw = w >> 2, gh = h >> 2;                         // the whole grid (scan indices)
  const int rw = w < 32 ? w : 32, rh = h < 32 ? h : 32;       // coefficients exist here
  const int rgw = rw >> 2, rgh = rh >> 2, nsb = rgw * rgh;
  const bool outside = gw * gh != nsb;                        // a 64-point side
  const int lw = rq_log2(w), lh = rq_log2(h), lrgw = rq_log2(rgw);
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
  const unsigned *cb = s.ctx_bits;
  int16_t *wl = s.wl;                     // the levels being decided, (rw + 2) x (rh + 2)
  const int WS = rw + 2;
  long long *fcs = rq_fcs(s);             // flag costs at / behind a scan position

  const int q4i = lane >> 2, j = lane & 3;   // quad of the group, row of the sub-block
  // the scan offsets of this lane's row, a nibble each (ascending with x)
  const unsigned krow = (unsigned)(kInv >> (16 * j)) & 0xffffu;
  auto kof = [krow](int i) { return (int)((krow >> (4 * i)) & 15u); };

  auto rec_pos = [&](int x, int y) {
    return ((y >> 2) * rgw + (x >> 2)) * RQ_SB_STRIDE + (((y & 3) << 2) | (x & 3));
  };
  auto quant = [&](int a) {  // GetFwdQuantFunc on a magnitude (rdo_quant.cc:949-964)
    return (int)(short)(int)((((long long)a * scale) + fq_offset) >> fq_shift);
  };
  auto dequant = [&](int lvl) {
    int deq;
    if (iq_shift > 0) deq = (lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift;
    else deq = (lvl * iq_scale) << -iq_shift;
    return (int)(short)d_clip3(deq, -32768, 32767);
  };
  auto err_of = [&](int abs_coeff, int level) {   // delta_u, rdo_quant.cc:360-365
    const long long orig_scaled =
        (((long long)abs_coeff * scale) + size_bias_offset) >> size_bias_shift;
    const long long quant_err = orig_scaled - ((long long)level << shift);
    return (int)(short)(quant_err >> (shift - 8));
  };
  // the template of decided neighbours (cabac.cc:535-552) of the coefficient at wl[p]
  auto neighbours = [&](int p, int &n_sig, int &n_g1, int &n_g2, int &sum_abs) {
    const int v0 = d_abs((int)wl[p + 1]), v1 = d_abs((int)wl[p + 2]),
              v2 = d_abs((int)wl[p + WS + 1]), v3 = d_abs((int)wl[p + WS]),
              v4 = d_abs((int)wl[p + 2 * WS]);
    n_sig = (v0 != 0) + (v1 != 0) + (v2 != 0) + (v3 != 0) + (v4 != 0);
    n_g1 = (v0 > 1) + (v1 > 1) + (v2 > 1) + (v3 > 1) + (v4 > 1);
    n_g2 = (v0 > 2) + (v1 > 2) + (v2 > 2) + (v3 > 2) + (v4 > 2);
    sum_abs = v0 + v1 + v2 + v3 + v4;
  };
  const int size_cls = (lw + lh) >> 1;
  const int sig_luma_size =
      size_cls > 2 && luma ? 18 << (size_cls - 3 < 1 ? size_cls - 3 : 1) : 0;
  auto sig_base_of = [&](int posxy) {  // GetCoeffSigCtx (cabac.cc:520-560) without the count
    int start = posxy < 2 ? 6 : 0;
    start += luma && posxy < 5 ? 6 : 0;
    return 2 * ((luma ? RQ_OFF(sig_luma) : RQ_OFF(sig_chroma)) + start + sig_luma_size);
  };
  auto sig_ctx_of = [&](int posxy, int n_sig) {
    return sig_base_of(posxy) + 2 * (n_sig < 5 ? n_sig : 5);
  };
  const int g1_off = luma ? RQ_OFF(greater1_luma) : RQ_OFF(greater1_chroma);
  auto greater_start_of = [&](int posxy) {   // cabac.cc:594-684
    return luma ? (posxy < 3 ? 10 : (posxy < 10 ? 5 : 0)) : 0;
  };
  auto greater_ctx_nn = [&](int posxy, int nn) {
    return nn == 0 ? 2 * g1_off : 2 * (g1_off + greater_start_of(posxy) + nn);
  };
  auto greater_nn = [](int n, bool is_last) { return is_last ? 0 : (n < 4 ? n : 4) + 1; };

  // ---- the units: magnitudes, plain quantised values, signs (two 16-bit halves a
  // register); per sub-block the zero distortion and the q != 0 set; the last position
  unsigned pa[NR][2], pq[NR][2], negs = 0, valid = 0;
  int scan_u[NR];
  long long zd_u[NR];          // (the sub-block's, equal on its four lanes)
  int last = -1;
  // zero levels + border, and "no sub-block of the corner at this scan position"
  for (int i = lane; i < (WS * (rh + 2)) >> 1; i += G) reinterpret_cast<uint32_t *>(wl)[i] = 0;
  if (outside)
    for (int i = lane; i < 64; i += G) reinterpret_cast<uint32_t *>(s.sb_of_scan)[i] = 0xffffffffu;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int sbi = UPR * r + q4i;
    const bool mine = sbi < nsb;
    const int sx = mine ? (sbi & (rgw - 1)) : 0, sy = mine ? (sbi >> lrgw) : 0;
    scan_u[r] = rq4_sb_scan_index(gw, gh, sx, sy);
    int c[4];
    if (mine) rq4_load_row(cf, sx << 2, (sy << 2) + j, c);
    else c[0] = c[1] = c[2] = c[3] = 0;
    unsigned qm = 0;
    unsigned long long sum_sq = 0;
    int a[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[i] = (short)d_abs(c[i]);
      q[i] = quant(a[i]);
      negs |= (unsigned)(c[i] < 0) << (4 * r + i);
      sum_sq += (unsigned)(a[i] * a[i]);
      if (q[i]) qm |= 1u << kof(i);
    }
    pa[r][0] = (unsigned)(a[0] & 0xffff) | ((unsigned)a[1] << 16);
    pa[r][1] = (unsigned)(a[2] & 0xffff) | ((unsigned)a[3] << 16);
    pq[r][0] = (unsigned)(q[0] & 0xffff) | ((unsigned)q[1] << 16);
    pq[r][1] = (unsigned)(q[2] & 0xffff) | ((unsigned)q[3] << 16);
    valid |= (unsigned)mine << r;
    const unsigned qmask = (unsigned)rq4_quad_or((int)qm);
    zd_u[r] = (long long)((unsigned long long)rq_group_sum_i64<4>((long long)sum_sq) << cost_scale);
    const int l = mine && qmask ? (scan_u[r] << 4) + 31 - __clz((int)qmask) : -1;
    last = l > last ? l : last;
  }
  const int last_pos_index = rq4_max<G>(last);
  RQ_TRACE(4);
  if (last_pos_index < 0) {  // nothing quantises to a level
#pragma unroll
    for (int r = 0; r < NR; r++)
      if ((valid >> r) & 1u) {
        const int sbi = UPR * r + q4i;
        const int z[4] = {0, 0, 0, 0};
        rq4_store_row(lev, (sbi & (rgw - 1)) << 2, ((sbi >> lrgw) << 2) + j, z);
      }
    return 0;
  }
  // unpack a unit's register pair (u: 0..3)
  auto half = [](const unsigned p[2], int u) {
    const unsigned w2 = u < 2 ? p[0] : p[1];
    return (int)(short)((u & 1) ? (w2 >> 16) : (w2 & 0xffffu));
  };

  const int last_k = last_pos_index & 15, last_sb = last_pos_index >> 4;
  int last_l = -1, d_first = -1;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int sbi = UPR * r + q4i;
    const bool mine = (valid >> r) & 1u;
    const bool live = mine && (scan_u[r] << 4) <= last_pos_index;
    if (mine && j == 0) {
      s.sb_of_scan[scan_u[r]] = (unsigned char)sbi;
      s.csbf[sbi] = 0;
      s.sb_dcz[sbi] = 0;
      s.sb_live[sbi] = live ? 1 : 0;
      if (!live) {
        s.csbf_bits[sbi] = 0;
        s.sb_code_cost[sbi] = zd_u[r];
      }
    }
    if (mine && scan_u[r] == last_sb) last_l = sbi;
    const int dd = live ? (sbi & (rgw - 1)) + (sbi >> lrgw) : -1;
    d_first = dd > d_first ? dd : d_first;
  }
  last_l = rq4_max<G>(last_l);
  d_first = rq4_max<G>(d_first);
  // The last-position bits (GetLastPosBits, rdo_quant.cc:918-965) of every position
  // GROUP of the two axes ([g] for x, [LPY + g] for y; diagonal scan: no swap): group g
  // costs the "1" bins of the groups in front of it - a prefix sum over the lanes, a
  // lane per group -, its own "0" bin unless it is the axis' last, and its suffix bits.
  constexpr int LPY = 16;
  {
    const int r16 = lane & 15;
    auto axis = [&](bool is_x, int &gmax) {
      gmax = rq_last_pos_group((is_x ? w : h) - 1);
      const bool inr = r16 < gmax;
      const int gc = gmax > 0 ? gmax - 1 : 0;
      const uint2 zb = *reinterpret_cast<const uint2 *>(
          cb + rq_last_pos_ctx(luma, w, h, r16 < gc ? r16 : gc, is_x));
      const unsigned one = inr ? zb.y : 0u, zero = inr ? zb.x : 0u;
      return rq4_row_scan_u32(one) - one + zero +
             (r16 > 3 ? (unsigned)((r16 - 2) >> 1) * RQ_BYPASS : 0u);
    };
    if (G == 64) {   // x on the wave's first row, y on the second
      int gmax;
      const unsigned v = axis(lane < 16, gmax);
      if (lane < 32 && r16 <= gmax) s.lp_bits[(lane < 16 ? 0 : LPY) + r16] = v;
    } else {         // (the groups of a wave share the table: same shape, same snapshot)
      int gx, gy;
      const unsigned vx = axis(true, gx), vy = axis(false, gy);
      if (r16 <= gx) s.lp_bits[r16] = vx;
      if (r16 <= gy) s.lp_bits[LPY + r16] = vy;
    }
  }
  // the two costs of a coefficient's significance flag as the decision used them
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
  wave_sync();
  RQ_TRACE(5);

  // ---- one anti-diagonal of sub-blocks at a time
  RQ_STEP_BEGIN();
  for (int d = d_first; d >= 0; d--) {
    // this lane's unit on the diagonal (at most one)
    int ru = 0;
    bool act = false;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int sbi = UPR * r + q4i;
      const bool on = ((valid >> r) & 1u) && (sbi & (rgw - 1)) + (sbi >> lrgw) == d &&
                      (scan_u[r] << 4) <= last_pos_index;
      if (on) {
        ru = r;
        act = true;
      }
    }
    unsigned ua[2], uq[2];
    int my_scan = scan_u[0];
    long long my_zero_dist = zd_u[0];
    ua[0] = pa[0][0]; ua[1] = pa[0][1]; uq[0] = pq[0][0]; uq[1] = pq[0][1];
#pragma unroll
    for (int r = 1; r < NR; r++)
      if (ru == r) {
        ua[0] = pa[r][0]; ua[1] = pa[r][1]; uq[0] = pq[r][0]; uq[1] = pq[r][1];
        my_scan = scan_u[r];
        my_zero_dist = zd_u[r];
      }
    const int sbi = UPR * ru + q4i;
    const int sx = sbi & (rgw - 1), sy = sbi >> lrgw;
    const int sb_index = my_scan << 4;
    const int row_p = ((sy << 2) + j) * WS + (sx << 2);   // wl index of the row's first level
    RQ_STEP(3);
    // step 1: the coefficients that have a choice (q != 0), along the sub-block's
    // anti-diagonals; c1 / c2 / nz = the sub-block's budget and count so far (equal
    // on its four lanes)
    int c1 = 0, c2 = 0, nz = 0;
    long long code_cost = 0;
    for (int sd = 6; sd >= 0; sd--) {
      const int x = sd - j;
      const bool on = act && (unsigned)x < 4u;
      const int xi = on ? x : 0;
      const int qv = half(uq, xi);
      const bool has = on && qv != 0;
      if (!__ballot(has)) continue;
      const int abs_coeff = half(ua, xi);
      const int k = kof(xi);
      const int p = row_p + xi;
      // the position class x + y is the step's: 4 d + sd
      const int posxy = 4 * d + sd;
      const int sig_base = sig_base_of(posxy), g_start = greater_start_of(posxy);
      // what does not depend on the budget
      int nn1 = 0, nn2 = 0, n_sig5 = 0;
      unsigned gr = 0, sig1 = 0;
      RdoqFlagBits fb = {0, 0, 0, 0};
      long long zero_alt = 0, dist_q = 0, dist_q1 = 0;
      bool zero_ok = false, dc_sig_zero = false;
      if (has) {
        const bool is_last = sb_index + k == last_pos_index;
        int n_sig, n_g1, n_g2, sum_abs;
        neighbours(p, n_sig, n_g1, n_g2, sum_abs);
        nn1 = greater_nn(n_g1, is_last);
        nn2 = greater_nn(n_g2, is_last);
        n_sig5 = n_sig < 5 ? n_sig : 5;
        {  // GetCoeffGolombRiceK (cabac.cc:686-725): smallest k with 2^(k+3) > threshold
          const unsigned threshold = 4u + (unsigned)(sum_abs - n_sig);
          const int kk = 29 - __clz((int)threshold);
          gr = (unsigned)(kk < 0 ? 0 : (kk > 9 ? 9 : kk));
        }
        const uint2 sig_b = *reinterpret_cast<const uint2 *>(cb + sig_base + 2 * n_sig5);
        const uint2 c1_b = *reinterpret_cast<const uint2 *>(
            cb + (nn1 == 0 ? 2 * g1_off : 2 * (g1_off + g_start + nn1)));
        const uint2 c2_b = *reinterpret_cast<const uint2 *>(
            cb + (nn2 == 0 ? 2 * g1_off : 2 * (g1_off + g_start + nn2)));
        fb.c1_0 = c1_b.x; fb.c1_1 = c1_b.y; fb.c2_0 = c2_b.x; fb.c2_1 = c2_b.y;
        // (k == 0 is alone on its anti-diagonal: nz is final for it)
        dc_sig_zero = sb_index > 0 && k == 0 && nz == 0;
        sig1 = (is_last || dc_sig_zero) ? 0u : sig_b.y;
        const long long zero_cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
        zero_ok = !is_last && qv < 3;
        zero_alt = zero_cost + rq_bit_cost(sig_b.x, lambda);
        if (qv > 0) {
          const int e0 = abs_coeff - dequant(qv);
          dist_q = ((long long)e0 * e0) << cost_scale;
          if (qv > 1) {
            const int e1 = abs_coeff - dequant(qv - 1);
            dist_q1 = ((long long)e1 * e1) << cost_scale;
          }
        }
      }
      // QuantCoeffRdo (rdo_quant.cc:708-735) + the zero alternative (:333-341) with the
      // budget as assumed; the quad's levels; again where the class was another
      int c1a = c1, c2a = c2;
      bool need = has;
      int best_level = 0;
      long long best_cost = 0;
      int v0, v1, v2, v3;
      for (;;) {
        if (need) {
          RdoqCoeffState st = {c1a, c2a, gr};
          best_cost = 0x7fffffffffffffffll;
          best_level = qv;   // (a magnitude of 32768 wraps to q < 0: no candidate but zero)
          if (qv > 0) {
            if (qv > 1) {
              best_cost = dist_q1 + rq_bit_cost(sig1 + rq_abs_level_bits(fb, qv - 1, st), lambda);
              best_level = qv - 1;
            }
            const long long cost =
                dist_q + rq_bit_cost(sig1 + rq_abs_level_bits(fb, qv, st), lambda);
            if (cost <= best_cost) {
              best_cost = cost;
              best_level = qv;
            }
          }
          if (zero_ok && zero_alt <= best_cost) {
            best_cost = zero_alt;
            best_level = 0;
          }
        }
        const int mylv = has ? best_level : 0;
        v0 = rq4_quad_bcast<0>(mylv);
        v1 = rq4_quad_bcast<1>(mylv);
        v2 = rq4_quad_bcast<2>(mylv);
        v3 = rq4_quad_bcast<3>(mylv);
        // reverse scan visits an anti-diagonal from its top-right end: rows ascending
        const int c1t = c1 + (j > 0 && v0 >= 1) + (j > 1 && v1 >= 1) + (j > 2 && v2 >= 1);
        const int c2t = c2 + (j > 0 && v0 >= 2) + (j > 1 && v1 >= 2) + (j > 2 && v2 >= 2);
        need = has && ((c1t < 8) != (c1a < 8) || (c2t < 1) != (c2a < 1));
        c1a = c1t;
        c2a = c2t;
        if (!__ballot(need)) break;
      }
      if (has) {
        wl[p] = (short)best_level;
        s.rate_up[rec_pos((sx << 2) + xi, (sy << 2) + j)] =
            RQ_STATE_PACK(nn1, nn2, c1a, c2a, gr, n_sig5);
        if (dc_sig_zero) s.sb_dcz[sbi] = 1;
        code_cost += best_cost;
      }
      c1 += (v0 >= 1) + (v1 >= 1) + (v2 >= 1) + (v3 >= 1);
      c2 += (v0 >= 2) + (v1 >= 2) + (v2 >= 2) + (v3 >= 2);
      nz += (v0 != 0) + (v1 != 0) + (v2 != 0) + (v3 != 0);
      wave_sync();   // the levels are in place for the next anti-diagonal's templates
    }
    code_cost = rq_group_sum_i64<4>(code_cost);
    bool any = nz != 0;
    if (act && j == 0) {
      s.sb_code_cost[sbi] = code_cost;
      // a sub-block without a level whose code cost EvalZeroSubblock discards
      // (:745-749): its coefficients without a choice need not be priced
      if (!any && !(sb_index == 0 || sb_index + 16 > last_pos_index)) s.sb_live[sbi] = 4;
    }
    wave_sync();
    RQ_STEP(0);
    // step 2: the coefficients without a choice of the diagonal's sub-blocks, one
    // per lane (sixteen lanes per sub-block)
    {
      const int ax0 = d > rgh - 1 ? d - (rgh - 1) : 0;
      const int ax1 = d < rgw - 1 ? d : rgw - 1;
      const int pairs = (ax1 - ax0 + 1) << 4;
      for (int t0 = 0; t0 < pairs; t0 += G) {
        const int t = t0 + lane;
        const bool in = t < pairs;
        const int ax = ax0 + ((in ? t : 0) >> 4), ay = d - ax, k = t & 15;
        const int l2 = ay * rgw + ax;
        const bool work = in && s.sb_live[l2] == 1;
        if (!__ballot(work)) continue;   // no sub-block of the round is priced at all
        long long cost = 0;
        if (work) {
          const int pp = rq_scan_pos(2, 0, k);
          const int x = (ax << 2) + (pp & 3), y = (ay << 2) + (pp >> 2);
          const int abs_coeff = (short)d_abs(cf(x, y));
          if (!quant(abs_coeff)) {  // (else: decided in step 1)
            cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
            if (l2 == last_l && k > last_k) {  // rdo_quant.cc:303-307 (+ the memsets :262-265)
              s.rate_up[rec_pos(x, y)] = (unsigned short)RQ_STATE_NO_RATE;
            } else {
              int n_sig, n_g1, n_g2, sum_abs;
              neighbours(y * WS + x, n_sig, n_g1, n_g2, sum_abs);
              cost += rq_bit_cost(cb[sig_ctx_of(x + y, n_sig)], lambda);
              s.rate_up[rec_pos(x, y)] = RQ_STATE_PACK(greater_nn(n_g1, false), 0, 0, 0, 0,
                                                       n_sig < 5 ? n_sig : 5);
            }
          }
        }
        cost = rq_group_sum_i64<16>(cost);
        if ((lane & 15) == 0 && in && cost) s.sb_code_cost[l2] += cost;
      }
    }
    wave_sync();
    RQ_STEP(1);
    // step 3: EvalZeroSubblock (rdo_quant.cc:737-775), the sub-block's first lane
    bool zeroed = false;
    if (act && j == 0) {
      long long sb_code_cost = s.sb_code_cost[sbi];
      const bool right = sx < rgw - 1 ? s.csbf[sbi + 1] != 0 : false;
      const bool below = sy < rgh - 1 ? s.csbf[sbi + rgw] != 0 : false;
      const int csbf_ctx = 2 * (RQ_OFF(csbf) + (luma ? 0 : 2) + ((right || below) ? 1 : 0));
      unsigned bits_to_zero = 0;
      bool zero_sb = false;
      if (!(sb_index == 0 || sb_index + 16 > last_pos_index)) {
        const unsigned z_bits = cb[csbf_ctx], c_bits = cb[csbf_ctx + 1];
        const long long zero_cost = my_zero_dist + rq_bit_cost(z_bits, lambda);
        if (any) {
          const long long cc = sb_code_cost + rq_bit_cost(c_bits, lambda);
          if (zero_cost < cc) {
            sb_code_cost = zero_cost;
            bits_to_zero = z_bits;
            zero_sb = true;
          } else {
            sb_code_cost = cc;
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
        s.sb_live[sbi] = 2;
      }
      s.csbf[sbi] = any ? 1 : 0;
      s.csbf_bits[sbi] = bits_to_zero;
      s.sb_code_cost[sbi] = sb_code_cost;
    }
    // a sub-block that was just zeroed: its rows' levels, by its four lanes
    if (__ballot(zeroed)) {
      const bool z = rq4_quad_or((int)zeroed) != 0;
      if (z) {
        wl[row_p] = 0; wl[row_p + 1] = 0; wl[row_p + 2] = 0; wl[row_p + 3] = 0;
      }
    }
    wave_sync();
    RQ_STEP(2);
  }
  RQ_STEP_END();
  long long code_part = 0, zero_part = 0;
#pragma unroll
  for (int r = 0; r < NR; r++)
    if (((valid >> r) & 1u) && j == 0) {
      code_part = rq4_add_wrap(code_part, s.sb_code_cost[UPR * r + q4i]);
      zero_part = rq4_add_wrap(zero_part, zd_u[r]);
    }
  long long comp_code_cost = rq4_sum_i64<G>(code_part);
  const long long comp_zero_dist = rq4_sum_i64<G>(zero_part);
  const unsigned outside_bits = cb[2 * (RQ_OFF(csbf) + (luma ? 0 : 2))];

  RQ_TRACE(6);
  // ---- EvalLastPos (rdo_quant.cc:777-850).  The reference walks back from the last
  // position carrying a running cost: minus every visited sub-block's flag cost, plus
  // cost_to_zero of every visited coefficient; a non-zero level is a candidate, the
  // first level above 1 ends the walk.  Here: the flag costs of the sub-blocks at or
  // behind a scan position as one suffix sum (a lane per scan position; a sub-block
  // beyond the corner costs a zero flag unless its coded flag is inferred); then only
  // the CODED sub-blocks between the last position's and the one that holds the highest
  // level above 1 are visited, sixteen lanes on the sixteen coefficients of one
  // sub-block, four sub-blocks per round with 64 lanes.
  int new_last = 0;
  const int cbf_ctx = 2 * (!luma ? RQ_OFF(cbf_chroma)
                                 : ((prm.flags & XVC_RDOQ_INTRA_CU) ? RQ_OFF(cbf_luma)
                                                                    : RQ_OFF(root_cbf)));
  const long long comp_zero_cost = comp_zero_dist + rq_bit_cost(cb[cbf_ctx], lambda);
  {
    RQ_STEP2_BEGIN();
    // the highest offset with a level above 1 among the coded sub-blocks of the walk
    int stop_local = -1;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int sbi = UPR * r + q4i;
      const bool mine = (valid >> r) & 1u;
      const bool coded = mine && scan_u[r] <= last_sb && s.csbf[mine ? sbi : 0] != 0;
      const int start_k = scan_u[r] == last_sb ? last_k : 15;
      const int rp = mine ? (((sbi >> lrgw) << 2) + j) * WS + ((sbi & (rgw - 1)) << 2) : 0;
      unsigned gt1 = ((unsigned)((int)wl[rp] > 1) << kof(0)) |
                     ((unsigned)((int)wl[rp + 1] > 1) << kof(1)) |
                     ((unsigned)((int)wl[rp + 2] > 1) << kof(2)) |
                     ((unsigned)((int)wl[rp + 3] > 1) << kof(3));
      gt1 = (unsigned)rq4_quad_or((int)gt1) & ((2u << start_k) - 1u);
      const int sl = coded && gt1 ? (scan_u[r] << 4) + 31 - __clz((int)gt1) : -1;
      stop_local = sl > stop_local ? sl : stop_local;
    }
    const int stop_idx = rq4_max<G>(stop_local);
    const int stop_sb = stop_idx >= 0 ? stop_idx >> 4 : 0;
    wave_sync();   // (sb_code_cost was read for the sums above: fcs may alias it)
    // by scan position: the flag cost summed over the positions at or behind it (up to
    // the last position's sub-block); the coded ones as bit sets
    const long long out_cost = rq_bit_cost(outside_bits, lambda);
    constexpr int RND = G == 64 ? 4 : 1;        // rounds of G scan positions (up to 256)
    unsigned long long todo[RND];
    long long carry = 0;                         // the rounds behind this one
    int n_out = 0;
#pragma unroll
    for (int rr = RND - 1; rr >= 0; rr--) {
      todo[rr] = 0;
      if (rr * G > last_sb) continue;            // (uniform)
      const int kk = rr * G + lane;
      const bool inw = kk <= last_sb && (outside || kk < nsb);
      const int tj = inw ? (outside ? (int)s.sb_of_scan[kk] : (int)s.sb_of_scan[kk]) : 255;
      const bool corner = inw && (!outside || tj != 255);
      long long fc = 0;
      if (corner) fc = rq_bit_cost(s.csbf_bits[tj], lambda);
      // beyond the corner: all zero, never the last one; a coded flag is spent on it
      // unless it is the scan's first sub-block (EvalZeroSubblock's csbf == 0 branch)
      const bool outc = inw && !corner && kk > 0 && kk < last_sb;
      if (outc) fc = out_cost;
      const bool cd = corner && kk >= stop_sb && s.csbf[corner ? tj : 0] != 0;
      const long long behind = rq4_add_wrap(rq4_suffix_i64<G>(fc), carry);
      if (inw) fcs[kk] = behind;
      if (G == 64) {
        carry = rq4_readlane_i64(behind, 0);
        todo[rr] = __ballot(cd);
        n_out += __popcll(__ballot(outc));
      } else {
        todo[rr] = (__ballot(cd) >> (ME2_LANE & 48)) & 0xffffull;
      }
    }
    // the sub-blocks beyond the corner in front of the last position cost their flag
    comp_code_cost = rq4_add_wrap(comp_code_cost, (long long)n_out * out_cost);
    wave_sync();
    RQ_STEP(0);
    const int kk = lane & 15;
    const int p16 = rq_scan_pos(2, 0, kk);
    const long long base = comp_code_cost + rq_bit_cost(cb[cbf_ctx + 1], lambda);
    long long best_cost = 0x7fffffffffffffffll;
    int best_last_plus1 = 0;
    long long acc = 0;                 // sum of the visited coded sub-blocks' cost_to_zero
    for (;;) {
      // this row's sub-block: the highest scan positions still to do, one per row
      int jj = -1;
      if (G == 64) {
        int pick[4] = {-1, -1, -1, -1};
        int n = 0;
#pragma unroll
        for (int rr = RND - 1; rr >= 0; rr--)
          while (n < 4 && todo[rr]) {
            const int b = 63 - __clzll((long long)todo[rr]);
            todo[rr] &= ~(1ull << b);
            pick[n++] = rr * 64 + b;
          }
        if (n == 0) break;
        const int row = lane >> 4;
        jj = row == 0 ? pick[0] : (row == 1 ? pick[1] : (row == 2 ? pick[2] : pick[3]));
      } else {
        if (!__ballot(todo[0] != 0)) break;
        if (todo[0]) {
          jj = 63 - __clzll((long long)todo[0]);
          todo[0] &= ~(1ull << jj);
        }
      }
      const bool on = jj >= 0;
      const int js = on ? jj : 0;
      const int t = on ? (int)s.sb_of_scan[js] : 0;
      const long long flags_behind = fcs[js];
      const bool dcz = s.sb_dcz[t] != 0;
      const int x = ((t & (rgw - 1)) << 2) + (p16 & 3), y = ((t >> lrgw) << 2) + (p16 >> 2);
      const int index = (js << 4) + kk;
      const int first_k = js == last_sb ? last_k : 15;
      const bool in = on && kk <= first_k && index >= stop_idx;
      const unsigned pk = (unsigned)s.rate_up[rec_pos(x, y)];
      const int v = (int)wl[y * WS + x];
      const int ac = (short)d_abs(cf(x, y));
      unsigned sig0, sig1;
      sig_pair(pk, x + y, index, kk, dcz, sig0, sig1);
      long long ctz = 0;
      if (in && index != stop_idx) {   // (the level that ends the walk adds nothing)
        ctz = -rq_bit_cost(sig0, lambda);
        if (v == 1) {
          // GetAbsLevelBits (rdo_quant.cc:852-896) for quant_level = 1
          const unsigned c1_0 = cb[greater_ctx_nn(x + y, (int)(pk & 7u))];
          const unsigned bits1 = sig1 + (((pk >> 6) & 1u) ? (2u + ((pk >> 8) & 15u)) * RQ_BYPASS
                                                         : RQ_BYPASS + c1_0);
          const int err = ac - dequant(1);
          ctz = (((long long)(ac * ac)) << cost_scale) -
                ((((long long)err * err) << cost_scale) + rq_bit_cost(bits1, lambda));
        } else if (v < 0) {
          // (a magnitude of 32768 decided as the last position: best_cost stayed at
          // its initial value)
          ctz = (((long long)(ac * ac)) << cost_scale) - 0x7fffffffffffffffll;
        }
      }
      const long long inc = rq_row_scan_i64(ctz);
      const long long total = rq_group_sum_i64<16>(ctz);
      long long acc_row = acc;         // + the rows in front of this one (higher positions)
      if (G == 64) {
        const long long t0 = rq4_readlane_i64(total, 0), t1 = rq4_readlane_i64(total, 16),
                        t2 = rq4_readlane_i64(total, 32), t3 = rq4_readlane_i64(total, 48);
        const int row = lane >> 4;
        const long long t01 = rq4_add_wrap(t0, t1), t012 = rq4_add_wrap(t01, t2);
        acc_row = rq4_add_wrap(acc_row, row == 0 ? 0ll : (row == 1 ? t0 : (row == 2 ? t01 : t012)));
        acc = rq4_add_wrap(acc, rq4_add_wrap(t012, t3));
      } else {
        acc = rq4_add_wrap(acc, total);
      }
      if (in && v != 0) {
        const unsigned lp_bits = s.lp_bits[rq_last_pos_group(x)] +
                                 s.lp_bits[LPY + rq_last_pos_group(y)];
        const long long cost = base - flags_behind + acc_row + (total - inc) +
                               rq_bit_cost(lp_bits, lambda) - rq_bit_cost(sig1, lambda);
        if (cost < best_cost) {        // (equal cost: the one met first, the higher index)
          best_cost = cost;
          best_last_plus1 = index + 1;
        }
      }
    }
    RQ_STEP(1);
    // the cheapest candidate (on equal cost the higher index), to every lane of the block
    auto keep_better = [&](long long oc, int oi) {
      if (oc < best_cost || (oc == best_cost && oi > best_last_plus1)) {
        best_cost = oc;
        best_last_plus1 = oi;
      }
    };
    keep_better(rq4_xor_i64<1>(best_cost), lane_xor<1>(best_last_plus1));
    keep_better(rq4_xor_i64<2>(best_cost), lane_xor<2>(best_last_plus1));
    {
      const int lo = rq4_row_half_mirror((int)(unsigned)best_cost),
                hi = rq4_row_half_mirror((int)(best_cost >> 32));
      keep_better((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo),
                  rq4_row_half_mirror(best_last_plus1));
    }
    {
      const int lo = rq4_row_mirror((int)(unsigned)best_cost),
                hi = rq4_row_mirror((int)(best_cost >> 32));
      keep_better((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo),
                  rq4_row_mirror(best_last_plus1));
    }
    if (G == 64) {
      const long long c0 = rq4_readlane_i64(best_cost, 0), c1r = rq4_readlane_i64(best_cost, 16),
                      c2r = rq4_readlane_i64(best_cost, 32), c3 = rq4_readlane_i64(best_cost, 48);
      const int i0 = __builtin_amdgcn_readlane(best_last_plus1, 0),
                i1 = __builtin_amdgcn_readlane(best_last_plus1, 16),
                i2 = __builtin_amdgcn_readlane(best_last_plus1, 32),
                i3 = __builtin_amdgcn_readlane(best_last_plus1, 48);
      best_cost = c0;
      best_last_plus1 = i0;
      keep_better(c1r, i1);
      keep_better(c2r, i2);
      keep_better(c3, i3);
    }
    new_last = comp_zero_cost < best_cost ? -1 : best_last_plus1;
    RQ_STEP(3);
    RQ_STEP2_END();
  }

  RQ_TRACE(7);
  // ---- zero what lies at / beyond the new last position, re-apply the signs: the
  // levels go out to the caller's array
  int nnz = 0;
  unsigned has_sb = 0;
  unsigned pl[NR][2];          // the unit's final levels, two per register
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int sbi = UPR * r + q4i;
    const bool mine = (valid >> r) & 1u;
    const int sx = sbi & (rgw - 1), sy = sbi >> lrgw;
    const int rp = mine ? ((sy << 2) + j) * WS + (sx << 2) : 0;
    int lvv[4] = {(int)wl[rp], (int)wl[rp + 1], (int)wl[rp + 2], (int)wl[rp + 3]};
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int level = lvv[i];
      if (new_last < 0 || (scan_u[r] << 4) + kof(i) >= new_last) level = 0;
      cnt += level != 0;
      lvv[i] = (short)((negs >> (4 * r + i)) & 1u ? -level : level);
    }
    if (!mine) cnt = 0;
    if (mine) rq4_store_row(lev, sx << 2, (sy << 2) + j, lvv);
    pl[r][0] = (unsigned)(lvv[0] & 0xffff) | ((unsigned)lvv[1] << 16);
    pl[r][1] = (unsigned)(lvv[2] & 0xffff) | ((unsigned)lvv[3] << 16);
    has_sb |= (unsigned)(rq4_quad_or(cnt) != 0) << r;
    nnz += cnt;
  }
  nnz = rq4_sum<G>(nnz);
  if (new_last < 0) return 0;
  if (!(sign_hide && nnz > 1)) return nnz;

  RQ_TRACE(8);
  // ---- CoeffSignHideRdo (rdo_quant.cc:575-705): four lanes per sub-block
  int lss = -1;
#pragma unroll
  for (int r = 0; r < NR; r++)
    if (((has_sb >> r) & 1u) && scan_u[r] > lss) lss = scan_u[r];
  const int last_sb_scan = rq4_max<G>(lss);
  int dn = 0;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int sbi = UPR * r + q4i;
    const int sx = sbi & (rgw - 1), sy = sbi >> lrgw;
    const int sb_index = scan_u[r] << 4;
    int lv4[4] = {half(pl[r], 0), half(pl[r], 1), half(pl[r], 2), half(pl[r], 3)};
    int first = 16, lastk = -1, sum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (lv4[i]) {
        const int k = kof(i);
        first = k < first ? k : first;
        lastk = k > lastk ? k : lastk;
        sum += lv4[i];
      }
    first = rq4_quad_min(first);
    lastk = rq4_quad_max(lastk);
    sum = dpp_group_sum<4>(sum);
    int fsl = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (lv4[i] && kof(i) == first) fsl = lv4[i] > 0 ? 0 : 1;
    const int first_sign = rq4_quad_or(fsl);
    const bool go = ((has_sb >> r) & 1u) && lastk - first >= 4 && first_sign != (sum & 1);
    if (!__ballot(go)) continue;
    const bool is_last_sb = scan_u[r] == last_sb_scan;
    const long long rd_factor = prm.rd_factor;
    const bool dcz = go ? s.sb_dcz[sbi] != 0 : false;
    long long best_cost = 0x7fffffffffffffffll;
    int best_delta = 0, best_k = -1;
    if (go) {
      const int k_top = is_last_sb ? lastk : 15;
      const int Y = (sy << 2) + j;
      // (the scan offsets of a row ascend with x: descending i visits them as the
      // reference does, the one met first wins on equal cost)
#pragma unroll
      for (int i = 3; i >= 0; i--) {
        const int k = kof(i);
        if (k > k_top) continue;
        const int X = (sx << 2) + i;
        const unsigned pk = (unsigned)s.rate_up[rec_pos(X, Y)];
        const int lvl = lv4[i];
        const bool ng = ((negs >> (4 * r + i)) & 1u) != 0;
        // delta_u of the level as it was decided (the sign was re-applied: undo it -
        // a magnitude of 32768 keeps its wrapped value)
        const int err_dist = err_of(half(pa[r], i), ng ? -lvl : lvl);
        long long cost;
        int delta;
        if (lvl != 0) {
          RdoqFlagBits fb;
          RdoqCoeffState st;
          state_of(pk, X + Y, fb, st);
          const int al = d_abs(lvl);
          const int lvl_rate = (int)rq_abs_level_bits(fb, al, st);
          const int rate_up = -lvl_rate + (int)rq_abs_level_bits(fb, al + 1, st);
          const int rate_down = -lvl_rate + (int)rq_abs_level_bits(fb, al - 1, st);
          const long long cost_inc = rd_factor * (-err_dist) + rate_up;
          long long cost_dec = rd_factor * err_dist + rate_down -
                               (al == 1 ? sig_rate_of(pk, X + Y, sb_index + k, k, dcz) : 0);
          if (is_last_sb && k == lastk && al == 1) cost_dec -= 4ll * RQ_BYPASS;
          if (cost_inc < cost_dec) {
            cost = cost_inc;
            delta = 1;
          } else {
            delta = -1;
            cost = (k == first && al == 1) ? 0x7fffffffll : cost_dec;
          }
        } else {
          const int rate0 =
              (pk & RQ_STATE_NO_RATE) ? 0 : (int)cb[greater_ctx_nn(X + Y, (int)(pk & 7u))];
          cost = rd_factor * -(long long)d_abs(err_dist) + rate0 +
                 sig_rate_of(pk, X + Y, sb_index + k, k, dcz) + (long long)RQ_BYPASS;
          delta = 1;
          if (k < first && (ng ? 1 : 0) != first_sign) cost = 0x7fffffffll;
        }
        if (cost < best_cost) {
          best_cost = cost;
          best_delta = delta;
          best_k = k;
        }
      }
    }
    // the sub-block's cheapest candidate: lowest cost, on equal cost the highest offset
    const int my_k = best_k;
    {
      long long oc = rq4_xor_i64<1>(best_cost);
      int ok = lane_xor<1>(best_k);
      if (oc < best_cost || (oc == best_cost && ok > best_k)) { best_cost = oc; best_k = ok; }
      oc = rq4_xor_i64<2>(best_cost);
      ok = lane_xor<2>(best_k);
      if (oc < best_cost || (oc == best_cost && ok > best_k)) { best_cost = oc; best_k = ok; }
    }
    if (go && my_k >= 0 && my_k == best_k) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (kof(i) == best_k) {
          const int before = lv4[i];
          if (before == 32767 || before == -32768) best_delta = -1;
          const int after =
              (short)(((negs >> (4 * r + i)) & 1u) ? before - best_delta : before + best_delta);
          *lev((sx << 2) + i, (sy << 2) + j) = (short)after;
          dn += (after != 0) - (before != 0);
        }
    }
  }
  return nnz + rq4_sum<G>(dn);
}

#endif  // XVCGPU_K_RDOQ4_H_
