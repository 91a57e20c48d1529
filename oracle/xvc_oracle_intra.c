/*
 * xvc_oracle_intra.c -- CPU restatement of xvc's intra prediction (67-mode
 * set, default restrictions) and of the SATD pre-selection pass of the intra
 * search.
 *
 * TEST INFRASTRUCTURE ONLY (see xvc_oracle.h).  Pinned against the
 * reference's IntraPrediction class and IntraSearch::DetermineSlowIntraModes
 * distortions through oracle/ref_harness.cc (tests/test_oracle_vs_ref.py).
 * Paths cited are relative to /root/reference/src.
 */
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"

#define RS XO_INTRA_REF_STRIDE /* kRefSampleStride_ = 2 * 64 + 1 */

/* intra_prediction.cc:38-50 (the 67-mode tables) */
static const int8_t kAngleExt[33] = {-32, -29, -26, -23, -21, -19, -17, -15, -13, -11, -9,
                                     -7,  -5,  -3,  -2,  -1,  0,   1,   2,   3,   5,   7,
                                     9,   11,  13,  15,  17,  19,  21,  23,  26,  29,  32};
static const int16_t kInvAngleExt[16] = {8192, 4096, 2731, 1638, 1170, 910, 745, 630,
                                         546,  482,  431,  390,  356,  315, 282, 256};

static int log2i(int v) {
  int n = 0;
  while ((1 << n) < v) n++;
  return n;
}

/* ComputeRefSamples (intra_prediction.cc:707-848) + FilterRefSamples
 * (:850-871).  `src` points at sample (0,0) of the block inside the
 * reconstruction plane.  ref[0] = corner, ref[1..w+h] = above + above-right,
 * ref[RS + 0..w+h-1] = left + below-left. */
void xo_intra_ref_samples(int bitdepth, int w, int h, int neighbors, int above_right,
                          int below_left, const uint16_t *src, ptrdiff_t stride,
                          uint16_t *ref, uint16_t *ref_filtered) {
  const uint16_t dc = (uint16_t)(1 << (bitdepth - 1));
  const int has_al = (neighbors & XVC_INTRA_HAS_ABOVE_LEFT) != 0;
  const int has_a = (neighbors & XVC_INTRA_HAS_ABOVE) != 0;
  const int has_l = (neighbors & XVC_INTRA_HAS_LEFT) != 0;
  const int tl = w, ls = w + h, ts = w + h;
  if (!has_al && !has_a && !has_l && !above_right && !below_left) {
    for (int x = 0; x < w + h + 1; x++) ref[x] = dc;
    for (int y = 0; y < h + w; y++) ref[RS + y] = dc;
  } else if (has_al && has_a && has_l && below_left == w && above_right == h) {
    const uint16_t *in = src - stride - 1;
    for (int x = 0; x < w + h + 1; x++) ref[x] = in[x];
    in += stride;
    for (int y = 0; y < h + w; y++) ref[RS + y] = in[y * stride];
  } else {
    uint16_t line[5 * 64];
    const int total = ls + ts + tl;
    for (int i = 0; i < total; i++) line[i] = dc;
    const uint16_t *st = src - stride - 1;
    uint16_t *lt = line + ls;
    if (has_al)
      for (int i = 0; i < tl; i++) lt[i] = st[0];
    st += stride;
    lt--;
    if (has_l) {
      for (int i = 0; i < h; i++) lt[-i] = st[i * stride];
      st += h * stride;
      lt -= h;
      if (below_left) {
        for (int i = 0; i < below_left; i++) lt[-i] = st[i * stride];
        for (int i = below_left; i < w; i++) lt[-i] = lt[-below_left + 1];
      }
    }
    st = src - stride;
    lt = line + ls + tl;
    if (has_a) {
      for (int i = 0; i < w; i++) lt[i] = st[i];
      if (above_right) {
        for (int i = 0; i < above_right; i++) lt[w + i] = st[w + i];
        for (int i = above_right; i < h; i++) lt[w + i] = lt[w + above_right - 1];
      }
    }
    /* reference padding (default: enabled) */
    if (!below_left) {
      uint16_t r;
      if (has_l) r = line[w];
      else if (has_al) r = line[ls];
      else if (has_a) r = line[ls + tl];
      else r = line[ls + tl + w];
      for (int i = 0; i < w; i++) line[i] = r;
    }
    if (!has_l)
      for (int i = 0; i < h; i++) line[w + i] = line[w - 1];
    if (!has_al)
      for (int i = 0; i < tl; i++) line[ls + i] = line[ls - 1];
    if (!has_a)
      for (int i = 0; i < w; i++) line[ls + tl + i] = line[ls + tl - 1];
    if (!above_right)
      for (int i = 0; i < h; i++) line[ls + tl + w + i] = line[ls + tl + w - 1];
    lt = line + ls + tl - 1;
    for (int x = 0; x < ts + 1; x++) ref[x] = lt[x];
    lt = line + ls - 1;
    for (int y = 0; y < ls; y++) ref[RS + y] = lt[-y];
  }
  if (!ref_filtered) return;
  const uint16_t *s = ref;
  uint16_t *d = ref_filtered;
  const int al = s[0];
  d[0] = (uint16_t)(((al << 1) + s[1] + s[RS] + 2) >> 2);
  for (int x = 1; x < w + h; x++) d[x] = (uint16_t)(((s[x] << 1) + s[x - 1] + s[x + 1] + 2) >> 2);
  d[w + h] = s[w + h];
  d[RS] = (uint16_t)(((s[RS] << 1) + al + s[RS + 1] + 2) >> 2);
  for (int y = 1; y < h + w; y++)
    d[RS + y] = (uint16_t)(((s[RS + y] << 1) + s[RS + y - 1] + s[RS + y + 1] + 2) >> 2);
  d[RS + h + w - 1] = s[RS + h + w - 1];
}

/* UseFilteredRefSamples (intra_prediction.cc:342-363), 67-mode thresholds */
int xo_intra_use_filtered(int w, int h, int mode) {
  static const int8_t thr[8] = {0, 20, 20, 14, 2, 0, 20, 0};
  const int size = (log2i(w) + log2i(h)) >> 1;
  const int dh = abs(mode - 18), dv = abs(mode - 50);
  return (dh < dv ? dh : dv) > thr[size];
}

static uint16_t clip_bd(int v, int max) { return (uint16_t)(v < 0 ? 0 : (v > max ? max : v)); }

/* AngularPred (intra_prediction.cc:425-558); the horizontal half works on
 * swapped references and is written transposed (the reference's flip-back). */
static void angular(int bitdepth, int w, int h, int mode, int filter, const uint16_t *ref,
                    uint16_t *out, ptrdiff_t os) {
  uint16_t flip[RS * 2];
  const int hor = mode < 34;
  const uint16_t *rp = ref;
  if (hor) {
    flip[0] = ref[0];
    for (int i = 0; i < w + h; i++) {
      flip[1 + i] = ref[RS + i];
      flip[RS + i] = ref[1 + i];
    }
    rp = flip;
    const int t = w;
    w = h;
    h = t;
  }
  const int angle_offset = hor ? 18 - mode : mode - 50;
  const int angle = kAngleExt[16 + angle_offset];
  const int max = (1 << bitdepth) - 1;
  /* tmp[y][x] in the (possibly swapped) frame */
  static __thread uint16_t tmp[64 * 64];
  if (!angle) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) tmp[y * 64 + x] = rp[1 + x];
    if (filter) {
      const int above_left = rp[0], above = rp[1];
      for (int y = 0; y < h; y++) {
        const int16_t v = (int16_t)(above + ((rp[RS + y] - above_left) >> 1));
        tmp[y * 64] = clip_bd(v, max);
      }
    }
  } else {
    uint16_t line_buf[RS];
    const uint16_t *line = rp + 1;
    if (angle < 0) {
      const int num_projected = -((h * angle) >> 5) - 1;
      uint16_t *base = line_buf + num_projected + 1;
      for (int i = 0; i < w + 1; i++) base[i - 1] = rp[i];
      const int inv = kInvAngleExt[-angle_offset - 1];
      int sum = 128;
      for (int i = 0; i < num_projected; i++) {
        sum += inv;
        base[-2 - i] = rp[RS + (sum >> 8) - 1];
      }
      line = base;
    }
    int asum = 0;
    for (int y = 0; y < h; y++) {
      asum += angle;
      const int off = asum >> 5, wt = asum & 31;
      for (int x = 0; x < w; x++)
        tmp[y * 64 + x] = wt ? (uint16_t)(((32 - wt) * line[off + x] + wt * line[off + x + 1] + 16) >> 5)
                             : line[off + x];
    }
    if (filter && abs(angle) <= 1)
      for (int y = 0; y < h; y++) {
        const int16_t v = (int16_t)(tmp[y * 64] + ((rp[RS + y] - rp[0]) >> 2));
        tmp[y * 64] = clip_bd(v, max);
      }
  }
  if (hor) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) out[x * os + y] = tmp[y * 64 + x];
  } else {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) out[y * os + x] = tmp[y * 64 + x];
  }
}

/* IntraPrediction::Predict (intra_prediction.cc:81-125) for planar, DC and the
 * 65 angular modes.  ref / ref_filtered as written by xo_intra_ref_samples. */
void xo_intra_predict(int bitdepth, int is_luma, int mode, int w, int h,
                      const uint16_t *ref, const uint16_t *ref_filtered, uint16_t *out,
                      ptrdiff_t os) {
  const uint16_t *r = (is_luma && xo_intra_use_filtered(w, h, mode)) ? ref_filtered : ref;
  const int post_filter = is_luma && w <= 16 && h <= 16;
  if (mode == 0) { /* PlanarPred :401-423 */
    const int wl = log2i(w), hl = log2i(h);
    const uint16_t *above = r + 1, *left = r + RS;
    const int top_right = r[1 + w], bottom_left = left[h];
    const int shift = wl + hl + 1, offset = 1 << (shift - 1);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const int hor = (h - 1 - y) * above[x] + (y + 1) * bottom_left;
        const int ver = (w - 1 - x) * left[y] + (x + 1) * top_right;
        out[y * os + x] = (uint16_t)(((hor << wl) + (ver << hl) + offset) >> shift);
      }
  } else if (mode == 1) { /* PredIntraDC :365-399, always the unfiltered references */
    int sum = 0;
    for (int x = 0; x < w; x++) sum += ref[1 + x];
    for (int y = 0; y < h; y++) sum += ref[RS + y];
    const int total = w + h;
    const uint16_t dc = (uint16_t)((sum + (total >> 1)) / total);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) out[y * os + x] = dc;
    if (post_filter) {
      for (int y = h - 1; y > 0; y--) out[y * os] = (uint16_t)((ref[RS + y] + 3 * dc + 2) >> 2);
      for (int x = 1; x < w; x++) out[x] = (uint16_t)((ref[1 + x] + 3 * dc + 2) >> 2);
      out[0] = (uint16_t)((ref[1] + ref[RS] + 2 * dc + 2) >> 2);
    }
  } else {
    angular(bitdepth, w, h, mode, post_filter, r, out, os);
  }
}

static int log2_floor(int x) { /* util::Log2Floor */
  int n = 0;
  while (x > 1) {
    n++;
    x >>= 1;
  }
  return n;
}

/* PredLmChroma (intra_prediction.cc:560-585) for 4:2:0: RescaleLuma (:873-906,
 * the CU's reconstructed luma and its row above / column to the left, [1 2 1;
 * 1 2 1] / 8), DeriveLmParams (:587-686, a least-squares line through the
 * neighbouring (down-scaled luma, chroma) pairs in integer arithmetic) and
 * AddLinearModel (sample_buffer.h:108-122).  (x, y, w, h) in chroma samples;
 * `luma` / `chroma` point at sample (0,0) of the reconstruction planes. */
void xo_intra_lm_chroma(int bitdepth, int x, int y, int w, int h, const uint16_t *luma,
                        ptrdiff_t ls, const uint16_t *chroma, ptrdiff_t cs, uint16_t *out,
                        ptrdiff_t os) {
  const int has_above = y > 0, has_left = x > 0;
  uint16_t sub_buf[(64 + 1) * (64 + 1)];
  const int ss = 65;
  uint16_t *sub = sub_buf + ss + 1; /* (0,0) of the block; row -1 / column -1 exist */
  const uint16_t *src0 = luma + (ptrdiff_t)(2 * y) * ls + 2 * x;
  const uint16_t *src = src0 + (has_above ? -2 * ls : 0);
  const int start_x = has_left ? 0 : 1, start_y = has_above ? -1 : 0;
  for (int yy = start_y; yy < h; yy++) {
    if (has_left) {
      const int sum = src[-3] + 2 * src[-2] + src[-1] + src[-3 + ls] + 2 * src[-2 + ls] + src[-1 + ls];
      sub[yy * ss - 1] = (uint16_t)((sum + 4) >> 3);
    } else {
      sub[yy * ss] = (uint16_t)((src[0] + src[ls] + 1) >> 1);
    }
    for (int xx = start_x; xx < w; xx++) {
      const int sum = src[2 * xx - 1] + 2 * src[2 * xx] + src[2 * xx + 1] + src[2 * xx - 1 + ls] +
                      2 * src[2 * xx + ls] + src[2 * xx + 1 + ls];
      sub[yy * ss + xx] = (uint16_t)((sum + 4) >> 3);
    }
    src += 2 * ls;
  }
  /* DeriveLmParams */
  int scale = 0, shift = 0, offset = 1 << (bitdepth - 1);
  if (has_above || has_left) {
    const uint16_t *cb = chroma + (ptrdiff_t)y * cs + x;
    int sum_x = 0, sum_y = 0, sum_xx = 0, sum_xy = 0, nbr = 0;
    if (has_above) {
      const int dx = has_left ? (w / h > 1 ? w / h : 1) : 1;
      for (int xx = 0; xx < w; xx += dx) {
        const int r = sub[-ss + xx], c = cb[-cs + xx];
        sum_x += r; sum_y += c; sum_xx += r * r; sum_xy += r * c; nbr++;
      }
    }
    if (has_left) {
      const int dy = has_above ? (h / w > 1 ? h / w : 1) : 1;
      for (int yy = 0; yy < h; yy += dy) {
        const int r = sub[yy * ss - 1], c = cb[yy * cs - 1];
        sum_x += r; sum_y += c; sum_xx += r * r; sum_xy += r * c; nbr++;
      }
    }
    int size_shift = 1; /* util::SizeToLog2 starts at 1 */
    while ((1 << size_shift) < nbr) size_shift++;
    if (size_shift > 15 - bitdepth) {
      const int sh = size_shift + bitdepth - 15;
      sum_x = (sum_x + (1 << (sh - 1))) >> sh;
      sum_y = (sum_y + (1 << (sh - 1))) >> sh;
      sum_xx = (sum_xx + (1 << (sh - 1))) >> sh;
      sum_xy = (sum_xy + (1 << (sh - 1))) >> sh;
      size_shift -= sh;
    }
    const int avg_x = sum_x >> size_shift, avg_y = sum_y >> size_shift;
    const int x_frac = sum_x & ((1 << size_shift) - 1), y_frac = sum_y & ((1 << size_shift) - 1);
    const int sxy = sum_xy - ((avg_x * avg_y) << size_shift) - (avg_x * y_frac) - (avg_y * x_frac);
    const int sxx = sum_xx - ((avg_x * avg_x) << size_shift) - 2 * avg_x * x_frac;
    int shift_xy = sxy == 0 ? 0 : log2_floor(abs(sxy)) - bitdepth + 2;
    if (shift_xy < 0) shift_xy = 0;
    int shift_xx = sxx == 0 ? 0 : log2_floor(abs(sxx)) - 5;
    if (shift_xx < 0) shift_xx = 0;
    const int sxy_s = sxy >> shift_xy, sxx_s = sxx >> shift_xx;
    const int total_shift = bitdepth + shift_xx + 4 + 7 - 13 - shift_xy;
    if (sxx_s < 32) {
      scale = 0; offset = avg_y; shift = 0;
    } else {
      int sc = (int)((uint32_t)sxy_s *
                     (uint32_t)(((1 << (bitdepth + 4)) + (sxx_s / 2)) / sxx_s));
      sc = sc >> total_shift;
      sc = sc < -256 ? -256 : (sc > 255 ? 255 : sc);
      scale = 128 * sc;
      const int base_shift = log2_floor(abs(scale) + (scale < 0 ? -1 : 0)) - (scale ? 5 : 0);
      shift = 13 - base_shift;
      scale >>= base_shift;
      offset = avg_y - ((scale * avg_x) >> shift);
    }
  }
  const int max = (1 << bitdepth) - 1;
  for (int yy = 0; yy < h; yy++)
    for (int xx = 0; xx < w; xx++) {
      const int v = ((scale * sub[yy * ss + xx]) >> shift) + offset;
      out[yy * os + xx] = (uint16_t)(v < 0 ? 0 : (v > max ? max : v));
    }
}

/* One job of the prediction batch: reference samples from the reconstruction
 * plane, then the block's prediction.  (XVC_INTRA_MODE_LM_CHROMA jobs need the
 * luma plane too: xo_intra_lm_chroma.) */
void xo_intra_pred_block(int bitdepth, const xvcgpu_intra_block *b, const uint16_t *rec,
                         ptrdiff_t rs, uint16_t *pred, ptrdiff_t ps) {
  uint16_t ref[RS * 2], filt[RS * 2];
  xo_intra_ref_samples(bitdepth, b->w, b->h, b->neighbors, b->above_right, b->below_left,
                       rec + (ptrdiff_t)b->y * rs + b->x, rs, ref, filt);
  xo_intra_predict(bitdepth, b->comp == 0, b->mode, b->w, b->h, ref, filt,
                   pred + (ptrdiff_t)b->y * ps + b->x, ps);
}

/* The distortion half of IntraSearch::DetermineSlowIntraModes
 * (xvc_enc_lib/intra_search.cc:189-305): SATD (CompareSample, weight 1) of the
 * original block against the prediction of every one of the 67 luma modes;
 * the host adds bits * lambda and folds. */
void xo_intra_satd_modes(int bitdepth, const xvcgpu_intra_block *b, const uint16_t *orig,
                         ptrdiff_t os, const uint16_t *rec, ptrdiff_t rs, uint32_t *dist) {
  uint16_t ref[RS * 2], filt[RS * 2], pred[64 * 64];
  xo_intra_ref_samples(bitdepth, b->w, b->h, b->neighbors, b->above_right, b->below_left,
                       rec + (ptrdiff_t)b->y * rs + b->x, rs, ref, filt);
  for (int m = 0; m < XVC_INTRA_NUM_MODES; m++) {
    xo_intra_predict(bitdepth, 1, m, b->w, b->h, ref, filt, pred, 64);
    dist[m] = (uint32_t)xo_metric_ss(XVC_METRIC_SATD, bitdepth, 0, 0, 1.0, b->w, b->h,
                                     orig + (ptrdiff_t)b->y * os + b->x, os, pred, 64);
  }
}
