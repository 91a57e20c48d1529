/*
 * xvc_oracle_rdoq.c -- CPU restatement of the reference's rate-distortion
 * optimised quantiser.  TEST INFRASTRUCTURE (the oracle).
 *
 * Restates  RdoQuant::QuantRdo<SubBlockShift>   xvc_enc_lib/rdo_quant.cc:223-446
 *           RdoQuant::QuantCoeffRdo             :689-720
 *           RdoQuant::EvalZeroSubblock          :722-760
 *           RdoQuant::EvalLastPos               :762-832
 *           RdoQuant::GetAbsLevelBits           :834-878
 *           RdoQuant::UpdateCodeState           :880-898
 *           RdoQuant::GetLastPosBits            :900-947
 *           RdoQuant::CoeffSignHideRdo          :575-687
 *           Get{Fwd,Inv}QuantFunc               :949-993
 * and the context selection it calls (extended residual context set, the
 * reference's default):
 *           CabacContexts::GetSubblockCsbfCtx   xvc_common_lib/cabac.cc:491-518
 *           GetCoeffSigCtx                      :520-592
 *           GetCoeffGreater1Ctx / Greater2Ctx   :594-684
 *           GetCoeffGolombRiceK                 :686-725
 *           GetCoeffLastPosCtx                  :727-770
 *           ContextModel::GetEntropyBits        xvc_common_lib/context_model.h:44-46
 * Pinned against RdoQuant::QuantRdo of the reference build by
 * tests/test_oracle_vs_ref.py (random and initialised context states).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"

/* ContextModel::kEntropyBits_ (context_model.cc:75-93): -log2 of the bin
 * probability of each of the 64 states x {MPS, LPS}, 15 fractional bits.  A
 * table of the CABAC engine's probability model; tests compare it entry by
 * entry with the reference's. */
static const uint32_t xq_entropy_bits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a,
    0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9, 0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3,
    0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600,
    0x03050, 0x10f95, 0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df,
    0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00, 0x01c99, 0x166de,
    0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547,
    0x0147c, 0x1a083, 0x0138e, 0x1a8a3, 0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b,
    0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d,
    0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577, 0x007c9, 0x24ce6, 0x00763, 0x25663,
    0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5,
    0x0055e, 0x29057, 0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f,
    0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb};
const uint32_t *xo_entropy_bits_table(void) { return xq_entropy_bits; }

/* Qp / Quantize helpers (quantize.cc:40-46, :58-63, :127-131) */
static const int xo_fwd_scales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
static const int xo_inv_scales[6] = {40, 45, 51, 57, 64, 72};
static int xo_clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int xo_qp_bitdepth(int qp_raw, int bd) {
  const int q = qp_raw + 6 * (bd - 8);
  return q > 0 ? q : 0;
}
static int xq_log2(int size);
static int xo_transform_shift(int w, int h, int bd) {
  return 15 - bd - ((xq_log2(w) + xq_log2(h)) >> 1);
}

#define XQ_BYPASS (1u << 15) /* ContextModel::kEntropyBypassBits */
#define XQ_LAMBDA_PREC 16    /* RdoQuant::kLambdaPrecision */

static uint32_t xq_bits(uint8_t state, int bin) { return xq_entropy_bits[state ^ bin]; }
static int64_t xq_bit_cost(uint32_t bits, int64_t lambda) {
  return ((int64_t)bits * lambda) >> XQ_LAMBDA_PREC;
}
static int xq_min(int a, int b) { return a < b ? a : b; }
static int xq_log2(int size) { /* util::SizeToLog2 */
  int l = 0;
  while ((1 << l) < size) l++;
  return l;
}

/* TransformHelper::kLastPosGroupIdx (transform.cc:47-55): group g covers
 * positions kLastPosMinInGroup[g] .. (next group's min) - 1 */
static int xq_last_pos_group(int pos) {
  static const uint8_t min_in_group[14] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96};
  int g = 13;
  while (min_in_group[g] > pos) g--;
  return g;
}
static const uint8_t xq_golomb_rice_range_ext[10] = {6, 5, 6, 3, 3, 3, 3, 3, 3, 3};
static const uint8_t xq_scan2x2[3][4] = {{0, 2, 1, 3}, {0, 1, 2, 3}, {0, 2, 1, 3}};
static const uint8_t xq_scan4x4[3][16] = {
    {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15}};

/* TransformHelper::DeriveSubblockScan (transform.cc:1639-1683) */
static void xq_subblock_scan(int order, int gw, int gh, uint16_t *tab) {
  int px = 0, py = 0;
  for (int i = 0; i < gw * gh; i++) {
    tab[i] = (uint16_t)(py * gw + px);
    if (order == 0) {
      if (px == gw - 1 || py == 0) {
        py += px + 1;
        px = 0;
        if (py >= gh) {
          px += py - (gh - 1);
          py = gh - 1;
        }
      } else {
        px++;
        py--;
      }
    } else if (order == 1) {
      if (px == gw - 1) {
        px = 0;
        py++;
      } else {
        px++;
      }
    } else {
      if (py == gh - 1) {
        px++;
        py = 0;
      } else {
        py++;
      }
    }
  }
}

typedef struct {
  int c1, c2, c1_idx, c2_idx;
  uint32_t golomb_rice_k;
} xq_state;

/* the 5-sample template to the right of / below a coefficient
 * (cabac.cc:535-552 and the same walk in :605-620, :651-666, :695-712) */
#define XQ_TEMPLATE(EXPR)                                       \
  do {                                                          \
    const int16_t *p_ = lv + (ptrdiff_t)y * ls + x;             \
    if (x < w - 1) {                                            \
      { const int v_ = p_[1]; EXPR; }                           \
      if (x < w - 2) { const int v_ = p_[2]; EXPR; }            \
      if (y < h - 1) { const int v_ = p_[1 + ls]; EXPR; }       \
    }                                                           \
    if (y < h - 1) {                                            \
      { const int v_ = p_[ls]; EXPR; }                          \
      if (y < h - 2) { const int v_ = p_[2 * ls]; EXPR; }       \
    }                                                           \
  } while (0)

static uint8_t xq_sig_ctx(const xvcgpu_rdoq_contexts *c, int luma, int x, int y,
                          const int16_t *lv, ptrdiff_t ls, int w, int h) {
  const int size = (xq_log2(w) + xq_log2(h)) >> 1, posxy = x + y;
  int offset = 0;
  XQ_TEMPLATE(offset += v_ != 0);
  offset = xq_min(offset, 5);
  int start = posxy < 2 ? 6 : 0;
  start += luma && posxy < 5 ? 6 : 0;
  start += size > 2 && luma ? 18 << xq_min(1, size - 3) : 0;
  return luma ? c->sig_luma[start + offset] : c->sig_chroma[start + offset];
}
static uint8_t xq_greater_ctx(const xvcgpu_rdoq_contexts *c, int luma, int thr, int x, int y,
                              int is_last, const int16_t *lv, ptrdiff_t ls, int w, int h) {
  if (is_last) return luma ? c->greater1_luma[0] : c->greater1_chroma[0];
  const int posxy = x + y;
  int offset = 0;
  XQ_TEMPLATE(offset += abs(v_) > thr);
  offset = xq_min(offset, 4) + 1;
  const int start = luma ? (posxy < 3 ? 10 : (posxy < 10 ? 5 : 0)) : 0;
  return luma ? c->greater1_luma[start + offset] : c->greater1_chroma[start + offset];
}
static uint32_t xq_golomb_rice_k(int x, int y, int w, int h, const int16_t *lv, ptrdiff_t ls) {
  int offset = 0, num = 0;
  XQ_TEMPLATE(offset += abs(v_); num += v_ != 0);
  const uint32_t threshold = 4 + (uint32_t)(offset - num);
  for (uint32_t k = 0; k < 10; k++)
    if ((1u << (k + 3)) > threshold) return k;
  return 9;
}
/* GetCoeffLastPosCtx (cabac.cc:727-770), alt_last_pos_ctx enabled */
static uint8_t xq_last_pos_ctx(const xvcgpu_rdoq_contexts *c, int luma, int w, int h, int pos,
                               int is_x) {
  static const uint8_t offset_ext[8] = {0, 0, 0, 3, 6, 10, 15, 21};
  const int size = is_x ? w : h;
  if (luma) {
    const int l2 = xq_log2(size);
    const int idx = offset_ext[l2] + (pos >> ((l2 + 1) >> 2));
    return is_x ? c->last_x_luma[idx] : c->last_y_luma[idx];
  }
  int shift = size >> 3;
  shift = shift < 0 ? 0 : (shift > 2 ? 2 : shift);
  return is_x ? c->last_x_chroma[pos >> shift] : c->last_y_chroma[pos >> shift];
}

static uint32_t xq_abs_level_bits(int level, uint8_t c1_ctx, uint8_t c2_ctx, const xq_state *s) {
  const int base_level = s->c1_idx < 8 ? (2 + (s->c2_idx < 1)) : 1;
  const uint32_t threshold = xq_golomb_rice_range_ext[s->golomb_rice_k];
  uint32_t bits = XQ_BYPASS; /* sign */
  if (level >= base_level) {
    uint32_t code = (uint32_t)(level - base_level);
    if (code < (threshold << s->golomb_rice_k)) {
      const int length = (int)(code >> s->golomb_rice_k);
      bits += (uint32_t)(length + 1 + (int)s->golomb_rice_k) * XQ_BYPASS;
    } else {
      int length = (int)s->golomb_rice_k;
      code -= threshold << s->golomb_rice_k;
      while (code >= (1u << length)) code -= 1u << (length++);
      const int num_bins = length + (int)threshold + length + 1 - (int)s->golomb_rice_k;
      bits += (uint32_t)num_bins * XQ_BYPASS;
    }
    if (s->c1_idx < 8) {
      bits += xq_bits(c1_ctx, 1);
      if (s->c2_idx < 1) bits += xq_bits(c2_ctx, 1);
    }
  } else if (level == 1) {
    bits += xq_bits(c1_ctx, 0);
  } else if (level == 2) {
    bits += xq_bits(c1_ctx, 1);
    bits += xq_bits(c2_ctx, 0);
  } else {
    return 0;
  }
  return bits;
}

static void xq_update_state(int level, xq_state *s) {
  const int base_level = s->c1_idx < 8 ? (2 + (s->c2_idx < 1)) : 1;
  if (level >= 1) s->c1_idx++;
  if (level >= 2) {
    s->c2_idx++;
    s->c1 = 0;
  } else if (level >= 1 && s->c1 < 3 && s->c1 > 0) {
    s->c1++;
  }
  if (level >= base_level && level > 3 * (1 << s->golomb_rice_k))
    s->golomb_rice_k = (uint32_t)xq_min((int)s->golomb_rice_k + 1, 4);
}

static uint32_t xq_last_pos_bits(const xvcgpu_rdoq_contexts *c, int luma, int w, int h,
                                 int scan_order, int lx, int ly) {
  if (scan_order == 2) {
    int t = lx; lx = ly; ly = t;
    t = w; w = h; h = t;
  }
  const int gx = xq_last_pos_group(lx), gy = xq_last_pos_group(ly);
  uint32_t bits = 0;
  int k;
  for (k = 0; k < gx; k++) bits += xq_bits(xq_last_pos_ctx(c, luma, w, h, k, 1), 1);
  if (gx < xq_last_pos_group(w - 1)) bits += xq_bits(xq_last_pos_ctx(c, luma, w, h, k, 1), 0);
  for (k = 0; k < gy; k++) bits += xq_bits(xq_last_pos_ctx(c, luma, w, h, k, 0), 1);
  if (gy < xq_last_pos_group(h - 1)) bits += xq_bits(xq_last_pos_ctx(c, luma, w, h, k, 0), 0);
  if (gx > 3) bits += (uint32_t)((gx - 2) >> 1) * XQ_BYPASS;
  if (gy > 3) bits += (uint32_t)((gy - 2) >> 1) * XQ_BYPASS;
  return bits;
}

/* RdoQuant::QuantRdo.  comp: 0 luma, else chroma.  in / out: w x h int16 with
 * strides is / os.  sign_hide: !disable_transform_sign_hiding.  Returns the
 * number of non-zero levels.  Per-thread scratch like the reference's member
 * arrays (rdo_quant.h:104-112). */
int xo_quant_rdo(int bd, int qp_raw, int comp, int scan_order, int sign_hide, int w, int h,
                 const xvcgpu_rdoq_contexts *ctx, const xvcgpu_rdoq_params *prm,
                 const int16_t *src, ptrdiff_t is, int16_t *out, ptrdiff_t os) {
  const int luma = comp == 0;
  if ((w == 2 || h == 2) && (prm->flags & XVC_RDOQ_NO_2X2)) /* :208-216 */
    return xo_quant_fast2(bd, qp_raw, 0, sign_hide, scan_order, w, h, src, is, out, os);
  const int sbs = (w == 2 || h == 2) ? 1 : 2; /* SubBlockShift */
  const int sb_size = 1 << (2 * sbs), sb_mask = (1 << sbs) - 1;
  const int gw = w >> sbs, gh = h >> sbs;
  const int qpb = xo_qp_bitdepth(qp_raw, bd);
  const int tshift = xo_transform_shift(w, h, bd);
  const int bias = ((xq_log2(w) + xq_log2(h)) & 1) != 0;
  const int shift = 14 + qpb / 6 + tshift;
  const int size_bias_shift = bias ? 7 : 0, size_bias_offset = bias ? 1 << 6 : 0;
  const int scale = xo_fwd_scales[qpb % 6] * (bias ? 181 : 1);
  const int cost_scale = 15 - 2 * tshift - 2 * (bd - 8) + 2 * bias;
  const int64_t lambda = prm->lambda;
  /* GetFwdQuantFunc / GetInvQuantFunc (:949-993) */
  const int fq_shift = shift + (bias ? 7 : 0);
  const int64_t fq_offset = (int64_t)1 << (fq_shift - 1);
  const int iq_shift = 6 - tshift + (bias ? 8 : 0);
  const int iq_scale = (xo_inv_scales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
  const uint8_t *cscan = sbs == 1 ? xq_scan2x2[scan_order] : xq_scan4x4[scan_order];

  static __thread uint16_t sb_scan[32 * 32];
  static __thread uint8_t sb_csbf[32 * 32];
  static __thread uint32_t csbf_bits_to_zero[32 * 32];
  static __thread int64_t cost_to_zero[64 * 64];
  static __thread uint32_t sig_bits_arr[64 * 64];
  static __thread int16_t err_dist[64 * 64];
  static __thread int sig_rate[64 * 64], rate_up[64 * 64], rate_down[64 * 64];
  xq_subblock_scan(scan_order, gw, gh, sb_scan);
  memset(sb_csbf, 0, (size_t)gw * gh);
  memset(err_dist, 0, sizeof(err_dist[0]) * w * h);
  memset(sig_rate, 0, sizeof(sig_rate[0]) * w * h);
  memset(rate_up, 0, sizeof(rate_up[0]) * w * h);
  memset(rate_down, 0, sizeof(rate_down[0]) * w * h);

  xq_state st = {1, 0, 0, 0, 0};
  int last_pos_index = -1;
  int64_t comp_zero_dist = 0, comp_code_cost = 0;

  for (int sbi = gw * gh - 1; sbi >= 0; sbi--) {
    const int sb_index = sbi << (2 * sbs);
    const int sb_pos = sb_scan[sbi];
    const int sb_y = sb_pos / gw, sb_x = sb_pos - sb_y * gw;
    const int last_c1 = st.c1;
    memset(&st, 0, sizeof(st));
    st.c1 = 1;
    (void)last_c1; /* ctx_set: only read by the non-extended context set */
    int64_t sb_zero_dist = 0, sb_code_cost = 0;
    /* GetSubblockCsbfCtx (cabac.cc:491-518) */
    const int right = sb_x < gw - 1 ? sb_csbf[sb_y * gw + sb_x + 1] != 0 : 0;
    const int below = sb_y < gh - 1 ? sb_csbf[(sb_y + 1) * gw + sb_x] != 0 : 0;
    const uint8_t csbf_ctx = ctx->csbf[luma ? 0 : 1][right | below];
    int num_non_zero = 0;

    for (int k = sb_size - 1; k >= 0; k--) {
      const int index = sb_index + k;
      const int x = (sb_x << sbs) + (cscan[k] & sb_mask), y = (sb_y << sbs) + (cscan[k] >> sbs);
      const int abs_coeff = (int16_t)abs(src[y * is + x]);
      const int64_t zero_cost = ((int64_t)(abs_coeff * abs_coeff)) << cost_scale;
      sb_zero_dist += zero_cost;
      const int q = (int16_t)(int)((((int64_t)abs_coeff * scale) + fq_offset) >> fq_shift);
      if (q && last_pos_index == -1) {
        last_pos_index = index;
      } else if (last_pos_index == -1) {
        out[y * os + x] = 0;
        sb_code_cost += zero_cost;
        continue;
      }
      const int is_last = index == last_pos_index;
      const uint8_t sig_ctx = xq_sig_ctx(ctx, luma, x, y, out, os, w, h);
      const uint8_t c1_ctx = xq_greater_ctx(ctx, luma, 1, x, y, is_last, out, os, w, h);
      const uint8_t c2_ctx = xq_greater_ctx(ctx, luma, 2, x, y, is_last, out, os, w, h);
      st.golomb_rice_k = xq_golomb_rice_k(x, y, w, h, out, os);
      const uint32_t sig0 = xq_bits(sig_ctx, 0);
      uint32_t sig1 = xq_bits(sig_ctx, 1);
      if (is_last || (sb_index > 0 && k == 0 && num_non_zero == 0)) sig1 = 0;

      int64_t best_cost = INT64_MAX;
      uint32_t best_sig = 0;
      int best_level = q;
      if (q > 0) { /* QuantCoeffRdo (:689-720) */
        best_sig = sig1;
        int64_t bc = INT64_MAX;
        int bl = q;
        for (int lvl = q > 1 ? q - 1 : q; lvl <= q; lvl++) {
          const uint32_t bits = sig1 + xq_abs_level_bits(lvl, c1_ctx, c2_ctx, &st);
          int deq;
          if (iq_shift > 0)
            deq = xo_clip3((lvl * iq_scale + (1 << (iq_shift - 1))) >> iq_shift, -32768, 32767);
          else
            deq = xo_clip3((lvl * iq_scale) << -iq_shift, -32768, 32767);
          const int64_t err = abs_coeff - (int16_t)deq;
          const int64_t cost = ((err * err) << cost_scale) + xq_bit_cost(bits, lambda);
          if (lvl == q - 1 || cost <= bc) {
            bc = cost;
            bl = lvl;
          }
        }
        best_cost = bc;
        best_level = bl;
      }
      if (!is_last && q < 3) {
        const int64_t cost = zero_cost + xq_bit_cost(sig0, lambda);
        if (cost <= best_cost) {
          best_cost = cost;
          best_sig = sig0;
          best_level = 0;
        }
      }
      out[y * os + x] = (int16_t)best_level;
      cost_to_zero[index] = zero_cost - best_cost;
      sig_bits_arr[index] = best_sig;
      sb_code_cost += best_cost;
      const int64_t orig_scaled = (((int64_t)abs_coeff * scale) + size_bias_offset) >> size_bias_shift;
      const int64_t quant_err = orig_scaled - ((int64_t)best_level << shift);
      err_dist[index] = (int16_t)(quant_err >> (shift - 8));
      sig_rate[index] = !is_last ? (int)(sig1 - sig0) : 0;
      if (best_level) {
        sb_csbf[sb_pos] = 1;
        num_non_zero++;
        const int lvl_rate = (int)xq_abs_level_bits(best_level, c1_ctx, c2_ctx, &st);
        rate_up[index] = -lvl_rate + (int)xq_abs_level_bits(best_level + 1, c1_ctx, c2_ctx, &st);
        rate_down[index] = -lvl_rate + (int)xq_abs_level_bits(best_level - 1, c1_ctx, c2_ctx, &st);
      } else {
        rate_up[index] = (int)xq_bits(c1_ctx, 0);
      }
      xq_update_state(best_level, &st);
    }

    /* EvalZeroSubblock (:722-760) */
    int zero_sb = 0;
    if (last_pos_index < 0 || sb_index == 0 || sb_index + sb_size > last_pos_index) {
      csbf_bits_to_zero[sb_pos] = 0;
    } else {
      const uint32_t z_bits = xq_bits(csbf_ctx, 0), c_bits = xq_bits(csbf_ctx, 1);
      const int64_t zero_cost = sb_zero_dist + xq_bit_cost(z_bits, lambda);
      if (sb_csbf[sb_pos]) {
        const int64_t code_cost = sb_code_cost + xq_bit_cost(c_bits, lambda);
        if (zero_cost < code_cost) {
          sb_code_cost = zero_cost;
          csbf_bits_to_zero[sb_pos] = z_bits;
          zero_sb = 1;
        } else {
          sb_code_cost = code_cost;
          csbf_bits_to_zero[sb_pos] = c_bits;
        }
      } else {
        sb_code_cost = zero_cost;
        csbf_bits_to_zero[sb_pos] = z_bits;
      }
    }
    if (zero_sb) {
      sb_csbf[sb_pos] = 0;
      for (int k = sb_size - 1; k >= 0; k--) {
        const int x = (sb_x << sbs) + (cscan[k] & sb_mask), y = (sb_y << sbs) + (cscan[k] >> sbs);
        out[y * os + x] = 0;
        cost_to_zero[sb_index + k] = 0;
      }
    }
    comp_code_cost += sb_code_cost;
    comp_zero_dist += sb_zero_dist;
  }
  if (last_pos_index < 0) return 0;

  /* EvalLastPos (:762-832) */
  {
    const uint8_t cbf_ctx = !luma ? ctx->cbf_chroma
                                  : ((prm->flags & XVC_RDOQ_INTRA_CU) ? ctx->cbf_luma : ctx->root_cbf);
    comp_code_cost += xq_bit_cost(xq_bits(cbf_ctx, 1), lambda);
    int start = last_pos_index % sb_size;
    int64_t best_cost = INT64_MAX;
    int best_last_plus1 = 0, stop = 0;
    for (int sbi = gw * gh - 1; sbi >= 0 && !stop; sbi--) {
      const int sb_index = sbi << (2 * sbs);
      if (sb_index > last_pos_index) continue;
      const int sb_pos = sb_scan[sbi];
      const int sb_y = sb_pos / gw, sb_x = sb_pos - sb_y * gw;
      comp_code_cost -= xq_bit_cost(csbf_bits_to_zero[sb_pos], lambda);
      if (!sb_csbf[sb_pos]) continue;
      for (int k = start; k >= 0; k--) {
        const int index = sb_index + k;
        const int x = (sb_x << sbs) + (cscan[k] & sb_mask), y = (sb_y << sbs) + (cscan[k] >> sbs);
        const int v = out[y * os + x];
        if (!v) {
          comp_code_cost += cost_to_zero[index];
          continue;
        }
        const uint32_t lp_bits = xq_last_pos_bits(ctx, luma, w, h, scan_order, x, y);
        const int64_t cost = comp_code_cost + xq_bit_cost(lp_bits, lambda) -
                             xq_bit_cost(sig_bits_arr[index], lambda);
        if (cost < best_cost) {
          best_cost = cost;
          best_last_plus1 = index + 1;
        }
        if (v > 1) {
          stop = 1;
          break;
        }
        comp_code_cost += cost_to_zero[index];
      }
      start = sb_size - 1;
    }
    const int64_t comp_zero_cost = comp_zero_dist + xq_bit_cost(xq_bits(cbf_ctx, 0), lambda);
    last_pos_index = comp_zero_cost < best_cost ? -1 : best_last_plus1;
  }
  if (last_pos_index < 0) return 0;

  /* zero out what lies beyond the new last position (:407-422).  NOTE the
   * reference passes the "plus 1" value on, so the coefficient AT
   * best_last_pos itself is the first one kept. */
  {
    const int last_sb_index = last_pos_index - (last_pos_index & (sb_size - 1));
    for (int sbi = gw * gh - 1; sbi >= 0; sbi--) {
      const int sb_index = sbi << (2 * sbs);
      if (sb_index < last_sb_index) break;
      const int sb_pos = sb_scan[sbi];
      const int sb_y = sb_pos / gw, sb_x = sb_pos - sb_y * gw;
      const int end = sb_index == last_sb_index ? last_pos_index % sb_size : 0;
      for (int k = sb_size - 1; k != end - 1; k--) {
        const int x = (sb_x << sbs) + (cscan[k] & sb_mask), y = (sb_y << sbs) + (cscan[k] >> sbs);
        out[y * os + x] = 0;
      }
    }
  }
  /* re-apply the signs */
  int nnz = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int level = out[y * os + x];
      nnz += level != 0;
      out[y * os + x] = (int16_t)(src[y * is + x] < 0 ? -level : level);
    }
  if (!(sign_hide && nnz > 1 && sbs > 1)) return nnz;

  /* CoeffSignHideRdo (:575-687) */
  const int64_t rd_factor = prm->rd_factor;
  nnz = 0;
  int is_last_sb = -1;
  for (int sbi = gw * gh - 1; sbi >= 0; sbi--) {
    const int sb_index = sbi << 4;
    const int sb_pos = sb_scan[sbi];
    const int sb_y = sb_pos / gw, sb_x = sb_pos - sb_y * gw;
#define XQ_POS(k) (((sb_y << 2) + (cscan[k] >> 2)) * os + (sb_x << 2) + (cscan[k] & 3))
#define XQ_SPOS(k) (((sb_y << 2) + (cscan[k] >> 2)) * is + (sb_x << 2) + (cscan[k] & 3))
    int first = 16, last = -1, sum = 0;
    for (int k = 15; k >= 0; k--) {
      const int v = out[XQ_POS(k)];
      if (v) {
        first = xq_min(first, k);
        last = last > k ? last : k;
        sum += v;
        nnz++;
      }
    }
    if (last >= 0 && is_last_sb == -1) is_last_sb = 1;
    if (last - first < 4) {
      if (is_last_sb == 1) is_last_sb = 0;
      continue;
    }
    const int first_sign = out[XQ_POS(first)] > 0 ? 0 : 1;
    if (first_sign == (sum & 1)) {
      if (is_last_sb == 1) is_last_sb = 0;
      continue;
    }
    int64_t best_cost = INT64_MAX;
    int best_delta = 0, best_k = -1;
    for (int k = is_last_sb == 1 ? last : 15; k >= 0; k--) {
      const int index = sb_index + k;
      const int lvl = out[XQ_POS(k)];
      int64_t cost;
      int delta;
      if (lvl != 0) {
        const int64_t cost_inc = rd_factor * (-err_dist[index]) + rate_up[index];
        int64_t cost_dec = rd_factor * err_dist[index] + rate_down[index] -
                           (abs(lvl) == 1 ? sig_rate[index] : 0);
        if (is_last_sb == 1 && k == last && abs(lvl) == 1) cost_dec -= 4 * (int64_t)XQ_BYPASS;
        if (cost_inc < cost_dec) {
          cost = cost_inc;
          delta = 1;
        } else {
          delta = -1;
          cost = (k == first && abs(lvl) == 1) ? INT32_MAX : cost_dec;
        }
      } else {
        cost = rd_factor * -(int64_t)abs(err_dist[index]) + rate_up[index] + sig_rate[index] +
               (int64_t)XQ_BYPASS;
        delta = 1;
        if (k < first && (src[XQ_SPOS(k)] >= 0 ? 0 : 1) != first_sign) cost = INT32_MAX;
      }
      if (cost < best_cost) {
        best_cost = cost;
        best_delta = delta;
        best_k = k;
      }
    }
    int16_t *o = &out[XQ_POS(best_k)];
    if (*o == 32767 || *o == -32768) best_delta = -1;
    if (!*o) nnz++;
    if (src[XQ_SPOS(best_k)] >= 0)
      *o = (int16_t)(*o + best_delta);
    else
      *o = (int16_t)(*o - best_delta);
    if (!*o) nnz--;
    if (is_last_sb == 1) is_last_sb = 0;
#undef XQ_POS
#undef XQ_SPOS
  }
  return nnz;
}

/* TransformEncoder::TransformAndReconstruct (transform_encoder.cc:203-285) with
 * the quantiser it really calls, QuantRdo (:230); blocks without XVC_TXF_RDOQ
 * go through xo_residual_pipeline (QuantFast).  Same conventions. */
int xo_residual_pipeline_rdoq(int bd, const xvcgpu_tx_block *b, const xvcgpu_rdoq_contexts *ctx,
                              const xvcgpu_rdoq_params *prm, const uint16_t *orig, ptrdiff_t os,
                              const uint16_t *pred, ptrdiff_t ps, uint16_t *rec, ptrdiff_t rs,
                              int16_t *coeff_out) {
  if (!(b->intra_pic & XVC_TXF_RDOQ))
    return xo_residual_pipeline(bd, b, orig, os, pred, ps, rec, rs, coeff_out);
  static __thread int16_t resi[64 * 64], coeff[64 * 64], deq[64 * 64];
  const int w = b->w, h = b->h;
  memset(resi, 0, sizeof(resi));
  const uint16_t *o = orig + (ptrdiff_t)b->y * os + b->x;
  const uint16_t *p = pred + (ptrdiff_t)b->y * ps + b->x;
  uint16_t *r = rec + (ptrdiff_t)b->y * rs + b->x;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) resi[y * 64 + x] = (int16_t)((int)o[y * os + x] - (int)p[y * ps + x]);
  const int skip = b->tx_hor == XVC_TX_SKIP;
  if (skip)
    xo_fwd_transform_skip(bd, w, h, resi, 64, coeff, 64);
  else
    xo_fwd_transform(bd, w, h, b->tx_hor, b->tx_ver, b->dst4x4, resi, 64, coeff, 64);
  const int nnz = xo_quant_rdo(bd, b->qp, b->comp, (b->intra_pic >> XVC_TXF_SCAN_SHIFT) & 3,
                               !(b->intra_pic & XVC_TXF_NO_SIGN_HIDING), w, h,
                               &ctx[prm->ctx_index], prm, coeff, 64, coeff_out, w);
  if (nnz) {
    const int dc_only = nnz == 1 && coeff_out[0] != 0;
    xo_dequant(bd, b->qp, w, h, coeff_out, w, deq, 64);
    if (skip)
      xo_inv_transform_skip(bd, w, h, deq, 64, resi, 64);
    else
      xo_inv_transform(bd, w, h, b->tx_hor, b->tx_ver, b->dst4x4, dc_only, deq, 64, resi, 64);
    const int smax = (1 << bd) - 1;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        r[y * rs + x] = (uint16_t)xo_clip3((int)p[y * ps + x] + resi[y * 64 + x], 0, smax);
  } else {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) r[y * rs + x] = p[y * ps + x];
  }
  return nnz;
}
