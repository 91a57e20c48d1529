// k_rdoq4.h -- RdoQuant::QuantRdo with FOUR lanes per 4x4 sub-block
// (xvc_enc_lib/rdo_quant.cc:224-446, QuantCoeffRdo :708-735, EvalZeroSubblock
// :737-775, EvalLastPos :777-850, CoeffSignHideRdo :575-705), for blocks whose
// coefficients are walked in the diagonal scan and fit a wave that way: 4x4 ..
// 32x32 with at most G / 4 sub-blocks (G = 16: up to 8x8; G = 64: up to 16x16,
// 8x32, 32x8).  Same arithmetic and the same per-coefficient records as
// wave_rdoq (k_rdoq.h: one lane per sub-block), which keeps the other scans, the
// 2-wide blocks and the blocks beyond sixteen sub-blocks.
//
// Why.  The launch of the packed quantiser lasts one wave's life, and a lone wave
// issues an instruction every 8 - 10 clocks whatever its lanes do: the life is the
// wave's DYNAMIC INSTRUCTION COUNT.  With a lane per sub-block every loop over a
// sub-block's sixteen coefficients (quantise + last position, zero-out and signs,
// the sign hiding's two passes) is sixteen trips, the coefficients without a
// choice of one anti-diagonal of sub-blocks take four rounds of sixteen lanes, and
// four blocks share a wave, so every data-dependent loop runs as often as the
// worst of four blocks needs.  Here lane = (sub-block, row of the sub-block):
//   * a lane holds its row's four coefficients, plain quantised values and scan
//     offsets in registers: the per-coefficient loops are four trips;
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
//     sub-blocks (at most 4 x 16) are one round, and no other block's trip counts
//     are paid for.
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
template <int G>
__device__ __forceinline__ long long rq4_sum_i64(long long v) {
  v = rq_group_sum_i64<16>(v);
  if (G == 64) {
    unsigned long long t = 0;
#pragma unroll
    for (int r = 0; r < 64; r += 16) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, r);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), r);
      t += ((unsigned long long)hi << 32) | lo;
    }
    v = (long long)t;
  }
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

// S: RdoqShared<N> or RdoqView (k_rdoq.h).  `lane` = 0..G-1, all G lanes call
// (G = 64: the whole wave; G = 16: four blocks of a wave side by side, w, h and
// the other scalar arguments then equal for the four).  s.ctx_bits holds the
// snapshot's bit costs (the caller staged them).  Returns the number of non-zero
// levels to every lane; every level of the block is written.
template <int G, typename S, typename CF, typename LEV>
__device__ __forceinline__ int wave_rdoq4(S &s, int lane, int bd, int w, int h, int comp_qp,
                                          bool luma, bool sign_hide,
                                          const xvcgpu_rdoq_params &prm, CF cf, LEV lev) {
  constexpr unsigned long long kInv = rq_pack_scan4_inv(0);   // y * 4 + x -> scan offset
  const int gw = w >> 2, gh = h >> 2, nsb = gw * gh;
  const int lw = rq_log2(w), lh = rq_log2(h), lgw = lw - 2;
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

  const int sbi = lane >> 2, j = lane & 3;
  const bool mine = sbi < nsb;
  const int sx = mine ? (sbi & (gw - 1)) : 0, sy = mine ? (sbi >> lgw) : 0;
  const int my_scan = d_sb_scan_index(0, gw, gh, sx, sy);
  const int sb_index = my_scan << 4;
  const int px = sx << 2, py = sy << 2, Y = py + j;
  // the scan offsets of this lane's row, a nibble each
  const unsigned krow = (unsigned)(kInv >> (16 * j)) & 0xffffu;

  auto rec_pos = [&](int x, int y) {
    return ((y >> 2) * gw + (x >> 2)) * RQ_SB_STRIDE + (((y & 3) << 2) | (x & 3));
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
  // the template of decided neighbours (cabac.cc:535-552)
  auto neighbours = [&](int x, int y, int &n_sig, int &n_g1, int &n_g2, int &sum_abs) {
    const int x1 = x + 1 < w ? x + 1 : x, x2 = x + 2 < w ? x + 2 : x;
    const int y1 = y + 1 < h ? y + 1 : y, y2 = y + 2 < h ? y + 2 : y;
    const int v0 = *lev(x1, y), v1 = *lev(x2, y), v2 = *lev(x1, y1), v3 = *lev(x, y1),
              v4 = *lev(x, y2);
    const bool m0 = x + 1 < w, m1 = x + 2 < w, m2 = m0 && y + 1 < h, m3 = y + 1 < h,
               m4 = y + 2 < h;
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
  auto greater_ctx_nn = [&](int posxy, int nn) {  // cabac.cc:594-684
    const int g1 = luma ? RQ_OFF(greater1_luma) : RQ_OFF(greater1_chroma);
    const int start = luma ? (posxy < 3 ? 10 : (posxy < 10 ? 5 : 0)) : 0;
    return nn == 0 ? 2 * g1 : 2 * (g1 + start + nn);
  };
  auto greater_nn = [](int n, bool is_last) { return is_last ? 0 : (n < 4 ? n : 4) + 1; };

  // ---- the row: magnitudes, plain quantised values, signs; zero distortion and the
  // q != 0 set of the sub-block; the last position
  int a4[4], q4[4];
  unsigned neg = 0, qm = 0;
  unsigned long long sum_sq = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = mine ? cf(px + i, Y) : 0;
    a4[i] = (short)d_abs(c);
    q4[i] = quant(a4[i]);
    neg |= (unsigned)(c < 0) << i;
    sum_sq += (unsigned)(a4[i] * a4[i]);
    if (q4[i]) qm |= 1u << ((krow >> (4 * i)) & 15u);
    if (mine) *lev(px + i, Y) = 0;
  }
  const unsigned qmask = (unsigned)rq4_quad_or((int)qm);
  const long long my_zero_dist = (long long)((unsigned long long)rq_group_sum_i64<4>((long long)sum_sq)
                                             << cost_scale);
  const int last = mine && qmask ? sb_index + 31 - __clz((int)qmask) : -1;
  const int last_pos_index = rq4_max<G>(last);
  RQ_TRACE(4);
  if (last_pos_index < 0) return 0;  // nothing quantises to a level

  const bool owner = mine && j == 0;   // the sub-block's per-sub-block records
  const bool live = mine && sb_index <= last_pos_index;
  if (owner) {
    s.sb_of_scan[my_scan] = (unsigned char)sbi;
    s.csbf[sbi] = 0;
    s.sb_dcz[sbi] = 0;
    s.sb_live[sbi] = live ? 1 : 0;
    if (!live) {
      s.csbf_bits[sbi] = 0;
      s.sb_code_cost[sbi] = my_zero_dist;
    }
  }
  const int last_k = last_pos_index & 15;
  const int last_l = rq4_max<G>(mine && (last_pos_index >> 4) == my_scan ? sbi : -1);
  const int d_first = rq4_max<G>(live ? sx + sy : -1);
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
    const bool act = live && sx + sy == d;
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
      const int qv = xi == 0 ? q4[0] : (xi == 1 ? q4[1] : (xi == 2 ? q4[2] : q4[3]));
      const bool has = on && qv != 0;
      if (!__ballot(has)) continue;
      const int abs_coeff = xi == 0 ? a4[0] : (xi == 1 ? a4[1] : (xi == 2 ? a4[2] : a4[3]));
      const int k = (int)((krow >> (4 * xi)) & 15u);
      const int X = px + xi;
      // what does not depend on the budget
      int nn1 = 0, nn2 = 0, n_sig5 = 0;
      unsigned gr = 0, sig1 = 0;
      RdoqFlagBits fb = {0, 0, 0, 0};
      long long zero_alt = 0, dist_q = 0, dist_q1 = 0;
      bool zero_ok = false, dc_sig_zero = false;
      if (has) {
        const int index = sb_index + k;
        const bool is_last = index == last_pos_index;
        int n_sig, n_g1, n_g2, sum_abs;
        neighbours(X, Y, n_sig, n_g1, n_g2, sum_abs);
        const int posxy = X + Y;
        nn1 = greater_nn(n_g1, is_last);
        nn2 = greater_nn(n_g2, is_last);
        n_sig5 = n_sig < 5 ? n_sig : 5;
        {  // GetCoeffGolombRiceK (cabac.cc:686-725): smallest k with 2^(k+3) > threshold
          const unsigned threshold = 4u + (unsigned)(sum_abs - n_sig);
          const int kk = 29 - __clz((int)threshold);
          gr = (unsigned)(kk < 0 ? 0 : (kk > 9 ? 9 : kk));
        }
        const uint2 sig_b = *reinterpret_cast<const uint2 *>(cb + sig_ctx_of(posxy, n_sig));
        const uint2 c1_b = *reinterpret_cast<const uint2 *>(cb + greater_ctx_nn(posxy, nn1));
        const uint2 c2_b = *reinterpret_cast<const uint2 *>(cb + greater_ctx_nn(posxy, nn2));
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
        *lev(X, Y) = (short)best_level;
        s.rate_up[rec_pos(X, Y)] = RQ_STATE_PACK(nn1, nn2, c1a, c2a, gr, n_sig5);
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
      const int ax0 = d > gh - 1 ? d - (gh - 1) : 0;
      const int ax1 = d < gw - 1 ? d : gw - 1;
      const int pairs = (ax1 - ax0 + 1) << 4;
      for (int t0 = 0; t0 < pairs; t0 += G) {
        const int t = t0 + lane;
        const bool in = t < pairs;
        const int ax = ax0 + ((in ? t : 0) >> 4), ay = d - ax, k = t & 15;
        const int l2 = ay * gw + ax;
        const bool work = in && s.sb_live[l2] == 1;
        if (!__ballot(work)) continue;   // no sub-block of the diagonal is priced at all
        long long cost = 0;
        if (work) {
          const int p = rq_scan_pos(2, 0, k);
          const int x = (ax << 2) + (p & 3), y = (ay << 2) + (p >> 2);
          const int pos = rec_pos(x, y);
          const int abs_coeff = (short)d_abs(cf(x, y));
          if (!quant(abs_coeff)) {  // (else: decided in step 1)
            cost = ((long long)(abs_coeff * abs_coeff)) << cost_scale;
            if (l2 == last_l && k > last_k) {  // rdo_quant.cc:303-307 (+ the memsets :262-265)
              s.rate_up[pos] = (unsigned short)RQ_STATE_NO_RATE;
            } else {
              int n_sig, n_g1, n_g2, sum_abs;
              neighbours(x, y, n_sig, n_g1, n_g2, sum_abs);
              cost += rq_bit_cost(cb[sig_ctx_of(x + y, n_sig)], lambda);
              s.rate_up[pos] = RQ_STATE_PACK(greater_nn(n_g1, false), 0, 0, 0, 0,
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
      const bool right = sx < gw - 1 ? s.csbf[sbi + 1] != 0 : false;
      const bool below = sy < gh - 1 ? s.csbf[sbi + gw] != 0 : false;
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
#pragma unroll
        for (int i = 0; i < 4; i++) *lev(px + i, Y) = 0;
      }
    }
    wave_sync();
    RQ_STEP(2);
  }
  RQ_STEP_END();
  const long long comp_code_cost = rq4_sum_i64<G>(owner ? s.sb_code_cost[sbi] : 0ll);
  const long long comp_zero_dist = rq4_sum_i64<G>(owner ? my_zero_dist : 0ll);

  RQ_TRACE(6);
  // ---- EvalLastPos (rdo_quant.cc:777-850).  The reference walks back from the last
  // position carrying a running cost: minus every visited sub-block's flag cost, plus
  // cost_to_zero of every visited coefficient; a non-zero level is a candidate, the
  // first level above 1 ends the walk.  Here: the flag costs of the sub-blocks at or
  // behind a scan position as one suffix sum (sixteen lanes, one per scan position);
  // then only the CODED sub-blocks between the last position's and the one that holds
  // the highest level above 1 are visited, sixteen lanes on the sixteen coefficients
  // of one sub-block per round (on real content one or two rounds; walking the
  // uncoded ones too was 4 of this section's 7 k clocks).
  int new_last = 0;
  const int cbf_ctx = 2 * (!luma ? RQ_OFF(cbf_chroma)
                                 : ((prm.flags & XVC_RDOQ_INTRA_CU) ? RQ_OFF(cbf_luma)
                                                                    : RQ_OFF(root_cbf)));
  const long long comp_zero_cost = comp_zero_dist + rq_bit_cost(cb[cbf_ctx], lambda);
  int lv4[4];   // this lane's row of levels (magnitudes)
  {
    RQ_STEP2_BEGIN();
    const int last_sb = last_pos_index >> 4;
    const bool visited = mine && my_scan <= last_sb;
    const bool coded = visited && s.csbf[sbi] != 0;
    const int start_k = my_scan == last_sb ? last_k : 15;
    // the highest offset with a level above 1, per sub-block
    unsigned gt1 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lv4[i] = mine ? (int)*lev(px + i, Y) : 0;
      gt1 |= (unsigned)(lv4[i] > 1) << ((krow >> (4 * i)) & 15u);
    }
    gt1 = (unsigned)rq4_quad_or((int)gt1) & ((2u << start_k) - 1u);
    const int stop_local = coded && gt1 ? sb_index + 31 - __clz((int)gt1) : -1;
    const int stop_idx = rq4_max<G>(stop_local);
    const int stop_sb = stop_idx >= 0 ? stop_idx >> 4 : 0;
    const int kk = lane & 15;
    const bool worker = lane < 16;
    // by scan position kk: the sub-block's flag cost, summed over the positions at or
    // behind it (up to the last position's sub-block); the coded ones as a bit set
    const int tj = kk < nsb ? (int)s.sb_of_scan[kk] : 0;
    const bool vis = worker && kk < nsb && kk <= last_sb;
    const long long fc = vis ? rq_bit_cost(s.csbf_bits[tj], lambda) : 0ll;
    const bool cd = vis && kk >= stop_sb && s.csbf[tj] != 0;
    const long long fc_behind = rq_group_sum_i64<16>(fc) - rq_row_scan_i64(fc) + fc;
    unsigned todo = (unsigned)((__ballot(cd) >> (ME2_LANE & ~(G - 1) & 63)) & 0xffffull);
    wave_sync();   // (sb_code_cost was read for the sums above)
    if (vis) s.sb_code_cost[kk] = fc_behind;
    wave_sync();
    RQ_STEP(0);
    const int p = rq_scan_pos(2, 0, kk);
    const long long base = comp_code_cost + rq_bit_cost(cb[cbf_ctx + 1], lambda);
    long long best_cost = 0x7fffffffffffffffll;
    int best_last_plus1 = 0;
    long long acc = 0;                 // sum of the visited coded sub-blocks' cost_to_zero
    while (__ballot(todo != 0)) {
      const bool on = todo != 0;
      const int jj = on ? 31 - __clz((int)todo) : 0;
      todo &= ~(1u << jj);
      const int t = (int)s.sb_of_scan[jj];
      const long long flags_behind = s.sb_code_cost[jj];
      const bool dcz = s.sb_dcz[t] != 0;
      const int x = ((t & (gw - 1)) << 2) + (p & 3), y = ((t >> lgw) << 2) + (p >> 2);
      const int index = (jj << 4) + kk;
      const int first_k = jj == last_sb ? last_k : 15;
      const bool in = on && worker && kk <= first_k && index >= stop_idx;
      const unsigned pk = (unsigned)s.rate_up[rec_pos(x, y)];
      const int v = (int)*lev(x, y);
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
      if (in && v != 0) {
        const unsigned lp_bits = s.lp_bits[rq_last_pos_group(x)] +
                                 s.lp_bits[LPY + rq_last_pos_group(y)];
        const long long cost = base - flags_behind + acc + (total - inc) +
                               rq_bit_cost(lp_bits, lambda) - rq_bit_cost(sig1, lambda);
        if (cost < best_cost) {        // (equal cost: the one met first, the higher index)
          best_cost = cost;
          best_last_plus1 = index + 1;
        }
      }
      acc += total;
    }
    RQ_STEP(1);
    // the cheapest candidate of the sixteen workers (on equal cost the higher index),
    // then to every lane of the block
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
    if (G == 64) {   // the workers are the wave's first row
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)best_cost, 0);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(best_cost >> 32), 0);
      best_cost = (long long)(((unsigned long long)hi << 32) | lo);
      best_last_plus1 = __builtin_amdgcn_readlane(best_last_plus1, 0);
    }
    new_last = comp_zero_cost < best_cost ? -1 : best_last_plus1;
    RQ_STEP(3);
    RQ_STEP2_END();
  }

  RQ_TRACE(7);
  // ---- zero what lies at / beyond the new last position, re-apply the signs
  int nnz = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int k = (int)((krow >> (4 * i)) & 15u);
    int level = lv4[i];
    if (new_last < 0 || sb_index + k >= new_last) level = 0;
    nnz += level != 0;
    lv4[i] = (short)((neg >> i) & 1u ? -level : level);
    if (mine) *lev(px + i, Y) = (short)lv4[i];
  }
  const bool has_sb = rq4_quad_or(nnz) != 0;
  nnz = rq4_sum<G>(nnz);
  if (new_last < 0) return 0;
  if (!(sign_hide && nnz > 1)) return nnz;

  RQ_TRACE(8);
  // ---- CoeffSignHideRdo (rdo_quant.cc:575-705): four lanes per sub-block
  const int last_sb_scan = rq4_max<G>(mine && has_sb ? my_scan : -1);
  int first = 16, lastk = -1, sum = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int k = (int)((krow >> (4 * i)) & 15u);
    if (lv4[i]) {
      first = k < first ? k : first;
      lastk = k > lastk ? k : lastk;
      sum += lv4[i];
    }
  }
  first = rq4_quad_min(first);
  lastk = rq4_quad_max(lastk);
  sum = dpp_group_sum<4>(sum);
  int fsl = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (lv4[i] && (int)((krow >> (4 * i)) & 15u) == first) fsl = lv4[i] > 0 ? 0 : 1;
  const int first_sign = rq4_quad_or(fsl);
  const bool go = mine && has_sb && lastk - first >= 4 && first_sign != (sum & 1);
  int dn = 0;
  if (__ballot(go)) {
    const bool is_last_sb = my_scan == last_sb_scan;
    const long long rd_factor = prm.rd_factor;
    const bool dcz = mine ? s.sb_dcz[sbi] != 0 : false;
    long long best_cost = 0x7fffffffffffffffll;
    int best_delta = 0, best_k = -1;
    if (go) {
      const int k_top = is_last_sb ? lastk : 15;
      // (the scan offsets of a row ascend with x: descending i visits them as the
      // reference does, the one met first wins on equal cost)
#pragma unroll
      for (int i = 3; i >= 0; i--) {
        const int k = (int)((krow >> (4 * i)) & 15u);
        if (k > k_top) continue;
        const int X = px + i;
        const unsigned pk = (unsigned)s.rate_up[rec_pos(X, Y)];
        const int lvl = lv4[i];
        const bool ng = ((neg >> i) & 1u) != 0;
        // delta_u of the level as it was decided (the sign was re-applied: undo it -
        // a magnitude of 32768 keeps its wrapped value)
        const int err_dist = err_of(a4[i], ng ? -lvl : lvl);
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
        if ((int)((krow >> (4 * i)) & 15u) == best_k) {
          const int before = lv4[i];
          if (before == 32767 || before == -32768) best_delta = -1;
          const int after = (short)(((neg >> i) & 1u) ? before - best_delta : before + best_delta);
          *lev(px + i, Y) = (short)after;
          dn = (after != 0) - (before != 0);
        }
    }
  }
  return nnz + rq4_sum<G>(dn);
}

#endif  // XVCGPU_K_RDOQ4_H_
