/*
 * xvc_oracle.c -- CPU restatement of the block-level primitives of the xvc hot
 * path: distortion metrics, interpolation, transforms, (de)quantisation.
 *
 * TEST INFRASTRUCTURE ONLY (see xvc_oracle.h).  Parity status: pinned against
 * oracle/_ref (the reference's own code) and tests/golden/.
 *
 * Plain C99, scalar, written from the arithmetic of the cited reference
 * functions; integer semantics are the int32 C semantics of the reference's
 * C kernels (not the saturating SSE2 variants - identical for <=12-bit
 * content, SURVEY.md section 7 hard part 5).
 * Build with -ffp-contract=off: a few metrics go through `double`.
 */
#include "xvc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define XO_MAX_BLK 64

static inline int xo_clip3(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}
static inline int xo_log2_size(int size) { /* util::SizeToLog2, utils.cc:29 */
  int l = 1;
  while ((1 << l) < size) l++;
  return l;
}
static inline uint16_t xo_clip_bd(int v, int max) { /* util::ClipBD, utils.h:53 */
  return (uint16_t)(v < 0 ? 0 : (v > max ? max : v));
}

/* ========================================================================= *
 *  Distortion metrics                                                       *
 * ========================================================================= */

/* One set of metric kernels per first-operand type (Sample / Residual). */
#define XO_DEFINE_METRICS(SFX, T1)                                            \
  /* ComputeSad_c, sample_metric.cc:670-684 */                                \
  static int xo_sad_##SFX(int w, int h, const T1 *a, ptrdiff_t sa,            \
                          const uint16_t *b, ptrdiff_t sb) {                  \
    int sum = 0;                                                              \
    for (int y = 0; y < h; y++) {                                             \
      for (int x = 0; x < w; x++) sum += abs((int)a[x] - (int)b[x]);          \
      a += sa;                                                                \
      b += sb;                                                                \
    }                                                                         \
    return sum;                                                               \
  }                                                                           \
  /* ComputeSsd_c, sample_metric.cc:300-314 */                                \
  static uint64_t xo_ssd_##SFX(int w, int h, const T1 *a, ptrdiff_t sa,       \
                               const uint16_t *b, ptrdiff_t sb) {             \
    uint64_t ssd = 0;                                                         \
    for (int y = 0; y < h; y++) {                                             \
      for (int x = 0; x < w; x++) {                                           \
        int d = (int)a[x] - (int)b[x];                                        \
        ssd += (uint64_t)(int64_t)(d * d);                                    \
      }                                                                       \
      a += sa;                                                                \
      b += sb;                                                                \
    }                                                                         \
    return ssd;                                                               \
  }                                                                           \
  /* CalcMeanDiff, sample_metric.cc:769-783 (C division truncates to 0) */    \
  static int xo_mean_diff_##SFX(int skip, int w, int h, const T1 *a,          \
                                ptrdiff_t sa, const uint16_t *b,              \
                                ptrdiff_t sb) {                               \
    int delta_sum = 0;                                                        \
    for (int y = 0; y < h; y += 1 + skip) {                                   \
      for (int x = 0; x < w; x++) delta_sum += (int)a[x] - (int)b[x];         \
      a += sa * (1 + skip);                                                   \
      b += sb * (1 + skip);                                                   \
    }                                                                         \
    return (delta_sum * (1 + skip)) / (w * h);                                \
  }                                                                           \
  /* ComputeSadAcOnly, sample_metric.cc:686-703 */                            \
  static uint64_t xo_sad_ac_##SFX(int skip, int bd, int w, int h,             \
                                  const T1 *a, ptrdiff_t sa,                  \
                                  const uint16_t *b, ptrdiff_t sb) {          \
    const int avg = xo_mean_diff_##SFX(skip, w, h, a, sa, b, sb);             \
    int sum = 0;                                                              \
    for (int y = 0; y < h; y += 1 + skip) {                                   \
      for (int x = 0; x < w; x++) sum += abs((int)a[x] - (int)b[x] - avg);    \
      a += sa * (1 + skip);                                                   \
      b += sb * (1 + skip);                                                   \
    }                                                                         \
    return (uint64_t)(int64_t)((sum * (1 + skip)) >> (bd - 8));               \
  }                                                                           \
  /* ComputeSatdNxM / ComputeSatd2x2, sample_metric.cc:403-668: Hadamard     \
   * H_th * D * H_tw of the (offset-removed) difference tile, sum |.|, then   \
   * the tile-size normalisation.  Sum of |.| is invariant to the butterfly   \
   * ordering so a plain in-place fast Walsh-Hadamard is bit-identical. */    \
  static int xo_satd_tile_##SFX(int tw, int th, const T1 *a, ptrdiff_t sa,    \
                                const uint16_t *b, ptrdiff_t sb, int off) {   \
    int m[16 * 16];                                                           \
    for (int y = 0; y < th; y++)                                              \
      for (int x = 0; x < tw; x++)                                            \
        m[y * tw + x] = (int)a[y * sa + x] - (int)b[y * sb + x] - off;        \
    for (int y = 0; y < th; y++) /* rows */                                   \
      for (int len = 1; len < tw; len <<= 1)                                  \
        for (int i = 0; i < tw; i += len << 1)                                \
          for (int j = i; j < i + len; j++) {                                 \
            int u = m[y * tw + j], v = m[y * tw + j + len];                   \
            m[y * tw + j] = u + v;                                            \
            m[y * tw + j + len] = u - v;                                      \
          }                                                                   \
    for (int x = 0; x < tw; x++) /* columns */                                \
      for (int len = 1; len < th; len <<= 1)                                  \
        for (int i = 0; i < th; i += len << 1)                                \
          for (int j = i; j < i + len; j++) {                                 \
            int u = m[j * tw + x], v = m[(j + len) * tw + x];                 \
            m[j * tw + x] = u + v;                                            \
            m[(j + len) * tw + x] = u - v;                                    \
          }                                                                   \
    int sum = 0;                                                              \
    for (int i = 0; i < tw * th; i++) sum += abs(m[i]);                       \
    if (tw == 2 && th == 2) return sum;             /* :643-668 */            \
    if (tw == 4 && th == 4) return (sum + 1) >> 1;  /* :632-633 */            \
    if (tw == th) return (sum + 2) >> 2;            /* :634-635 */            \
    return (int)(2.0 * sum / sqrt((double)(tw * th))); /* :637-638 */         \
  }                                                                           \
  /* ComputeSatd tile selection, sample_metric.cc:316-389 */                  \
  static uint64_t xo_satd_##SFX(int bd, int w, int h, int off, const T1 *a,   \
                                ptrdiff_t sa, const uint16_t *b,              \
                                ptrdiff_t sb) {                               \
    int tw, th;                                                               \
    if (w == 2 || h == 2) {                                                   \
      tw = 2; th = 2;                                                         \
    } else if (w == 4 && h == 4) {                                            \
      tw = 4; th = 4;                                                         \
    } else if (h == 4 && w > h) {                                             \
      tw = 8; th = 4;                                                         \
    } else if (w == 4 && h > w) {                                             \
      tw = 4; th = 8;                                                         \
    } else if (w > h) {                                                       \
      tw = 16; th = 8;                                                        \
    } else if (w < h) {                                                       \
      tw = 8; th = 16;                                                        \
    } else {                                                                  \
      tw = 8; th = 8;                                                         \
    }                                                                         \
    uint64_t sad = 0;                                                         \
    for (int y = 0; y < h; y += th)                                           \
      for (int x = 0; x < w; x += tw)                                         \
        sad += (uint64_t)(int64_t)xo_satd_tile_##SFX(                         \
            tw, th, a + y * sa + x, sa, b + y * sb + x, sb, off);             \
    return sad >> (bd - 8);                                                   \
  }                                                                           \
  /* ComputeStructuralSsdBlock, sample_metric.cc:705-748 */                   \
  static uint64_t xo_sssd_block_##SFX(int bd, int qp_raw, int strength,       \
                                      int size, const T1 *a, ptrdiff_t sa,    \
                                      const uint16_t *b, ptrdiff_t sb) {      \
    int64_t orig_sum = 0, reco_sum = 0, oo = 0, rr = 0, orr = 0, ssd = 0;     \
    const int n = size * size;                                                \
    const int shift = 2 * (bd - 8);                                           \
    const int64_t c1 =                                                        \
        (int64_t)(((unsigned long long)(n * n) * 26634ull >> 12) << shift);   \
    const int64_t c2 =                                                        \
        (int64_t)(((unsigned long long)(n * n) * 239708ull >> 12) << shift);  \
    const int64_t c4 = (int64_t)(((1ull << 8) - 1) * ((1 << 8) - 1));         \
    const int z = qp_raw;                                                     \
    int wtmp = (int)((4 * z - 0.054 * z * z - 70) * strength);                \
    const int w = (wtmp > 0 ? wtmp : 0) >> 4;                                 \
    const int w1 = 64 - (w >> 1);                                             \
    const int w2 = 2 * w;                                                     \
    for (int y = 0; y < size; y++) {                                          \
      for (int x = 0; x < size; x++) {                                        \
        int o = (int)a[x], r = (int)b[x];                                     \
        orig_sum += o;                                                        \
        reco_sum += r;                                                        \
        oo += o * o;                                                          \
        rr += r * r;                                                          \
        orr += o * r;                                                         \
        int d = o - r;                                                        \
        ssd += d * d;                                                         \
      }                                                                       \
      a += sa;                                                                \
      b += sb;                                                                \
    }                                                                         \
    double m = (1.0 * orig_sum - reco_sum) / n;                               \
    double aa = (c4 - m * m + c1) / (c4 + c1);                                \
    double bb = (2.0 * n * orr - 2 * orig_sum * reco_sum + c2) /              \
                (n * oo - orig_sum * orig_sum + n * rr - reco_sum * reco_sum + \
                 c2);                                                         \
    ssd >>= shift;                                                            \
    return (uint64_t)(w1 * ssd +                                              \
                      w2 * (c4 >> ((8 - size) >> 1)) * (1 - aa * bb)) >>      \
           6;                                                                 \
  }                                                                           \
  /* ComputeStructuralSsd, sample_metric.cc:750-767 */                        \
  static uint64_t xo_sssd_##SFX(int bd, int qp_raw, int strength, int w,      \
                                int h, const T1 *a, ptrdiff_t sa,             \
                                const uint16_t *b, ptrdiff_t sb) {            \
    int size = (h < 8 || w < 8) ? 4 : 8;                                      \
    uint64_t ssim = 0;                                                        \
    for (int i = 0; i < h / size; i++) {                                      \
      for (int j = 0; j < w / size; j++)                                      \
        ssim += xo_sssd_block_##SFX(bd, qp_raw, strength, size, a + size * j, \
                                    sa, b + size * j, sb);                    \
      a += size * sa;                                                         \
      b += size * sb;                                                         \
    }                                                                         \
    return ssim;                                                              \
  }                                                                           \
  /* SampleMetric::Compare, sample_metric.cc:171-277 */                       \
  static uint64_t xo_compare_##SFX(int metric, int bd, int qp_raw,            \
                                   int strength, double weight, int w, int h, \
                                   const T1 *a, ptrdiff_t sa,                 \
                                   const uint16_t *b, ptrdiff_t sb) {         \
    uint64_t dist;                                                            \
    switch (metric) {                                                         \
      case XVC_METRIC_SSD:                                                    \
        dist = xo_ssd_##SFX(w, h, a, sa, b, sb) >> (2 * (bd - 8));            \
        break;                                                                \
      case XVC_METRIC_SATD:                                                   \
        dist = xo_satd_##SFX(bd, w, h, 0, a, sa, b, sb);                      \
        break;                                                                \
      case XVC_METRIC_SATD_ACONLY:                                            \
        dist = xo_satd_##SFX(bd, w, h,                                        \
                             xo_mean_diff_##SFX(0, w, h, a, sa, b, sb), a,    \
                             sa, b, sb);                                      \
        break;                                                                \
      case XVC_METRIC_SAD:                                                    \
        dist = (uint64_t)(int64_t)xo_sad_##SFX(w, h, a, sa, b, sb);           \
        dist = dist >> (bd - 8);                                              \
        break;                                                                \
      case XVC_METRIC_SAD_FAST:                                               \
        dist = (uint64_t)(int64_t)xo_sad_##SFX(w, h / 2, a, sa * 2, b,        \
                                               sb * 2);                       \
        dist = (dist * 2) >> (bd - 8);                                        \
        break;                                                                \
      case XVC_METRIC_SAD_ACONLY:                                             \
        dist = xo_sad_ac_##SFX(0, bd, w, h, a, sa, b, sb);                    \
        break;                                                                \
      case XVC_METRIC_SAD_ACONLY_FAST:                                        \
        dist = xo_sad_ac_##SFX(1, bd, w, h, a, sa, b, sb);                    \
        break;                                                                \
      case XVC_METRIC_STRUCTURAL_SSD:                                         \
        dist = xo_sssd_##SFX(bd, qp_raw, strength, w, h, a, sa, b, sb);       \
        break;                                                                \
      default:                                                                \
        return UINT64_MAX;                                                    \
    }                                                                         \
    return (uint64_t)((double)dist * weight);                                 \
  }

XO_DEFINE_METRICS(ss, uint16_t)
XO_DEFINE_METRICS(rs, int16_t)

uint64_t xo_metric_ss(int metric, int bitdepth, int qp_raw_y, int strength,
                      double weight, int w, int h, const uint16_t *s1,
                      ptrdiff_t st1, const uint16_t *s2, ptrdiff_t st2) {
  return xo_compare_ss(metric, bitdepth, qp_raw_y, strength, weight, w, h, s1,
                       st1, s2, st2);
}

uint64_t xo_metric_rs(int metric, int bitdepth, int qp_raw_y, int strength,
                      double weight, int w, int h, const int16_t *s1,
                      ptrdiff_t st1, const uint16_t *s2, ptrdiff_t st2) {
  return xo_compare_rs(metric, bitdepth, qp_raw_y, strength, weight, w, h, s1,
                       st1, s2, st2);
}

uint64_t xo_ssd_rr(int bitdepth, double weight, int w, int h,
                   const int16_t *s1, ptrdiff_t st1, const int16_t *s2,
                   ptrdiff_t st2) {
  uint64_t ssd = 0;
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      int d = (int)s1[x] - (int)s2[x];
      ssd += (uint64_t)(int64_t)(d * d);
    }
    s1 += st1;
    s2 += st2;
  }
  ssd >>= 2 * (bitdepth - 8);
  return (uint64_t)((double)ssd * weight);
}

/* ComparePicture / ComputePsnr block walk, sample_metric.cc:37-155.
 * NOTE the reference's loop bounds: full 64x64 blocks are visited only while
 * x < width - 64 / y < height - 64 (strict), the remainder in min_block steps
 * where min_block = lowest set bit of the dimension. */
uint64_t xo_picture_ssd(int bitdepth, int w, int h, const uint16_t *p1,
                        ptrdiff_t st1, const uint16_t *p2, ptrdiff_t st2,
                        uint64_t *psnr_dist, uint64_t *psnr_samples) {
  return xo_picture_ssd_rows(bitdepth, w, h, 0, h, p1, st1, p2, st2, psnr_dist,
                             psnr_samples);
}

/* The same walk restricted to the blocks whose first row lies in
 * [y_begin, y_end): the share of one CU-row shard (the shares of disjoint row
 * ranges covering the picture add up to xo_picture_ssd). */
uint64_t xo_picture_ssd_rows(int bitdepth, int w, int h, int y_begin, int y_end,
                             const uint16_t *p1, ptrdiff_t st1, const uint16_t *p2,
                             ptrdiff_t st2, uint64_t *psnr_dist,
                             uint64_t *psnr_samples) {
  const int B = 64;
  const int mbx = w & ~(w - 1);
  const int mby = h & ~(h - 1);
  const int sh = 2 * (bitdepth - 8);
  uint64_t dist = 0, samples = 0;
  int y;
  for (y = 0; y < h - B; y += B) {
    const int take = y >= y_begin && y < y_end;
    for (int x = 0; take && x < w - B; x += B) {
      dist += xo_ssd_ss(B, B, p1 + x, st1, p2 + x, st2) >> sh;
      samples += (uint64_t)B * B;
    }
    for (int x = w & ~(B - 1); take && x < w; x += mbx) {
      dist += xo_ssd_ss(mbx, B, p1 + x, st1, p2 + x, st2) >> sh;
      samples += (uint64_t)mbx * B;
    }
    p1 += st1 * B;
    p2 += st2 * B;
  }
  for (y = h & ~(B - 1); y < h; y += mby) {
    const int take = y >= y_begin && y < y_end;
    for (int x = 0; take && x < w - B; x += B) {
      dist += xo_ssd_ss(B, mby, p1 + x, st1, p2 + x, st2) >> sh;
      samples += (uint64_t)B * mby;
    }
    for (int x = w & ~(B - 1); take && x < w; x += mbx) {
      dist += xo_ssd_ss(mbx, mby, p1 + x, st1, p2 + x, st2) >> sh;
      samples += (uint64_t)mbx * mby;
    }
    p1 += st1 * mby;
    p2 += st2 * mby;
  }
  if (psnr_dist) *psnr_dist = dist;
  if (psnr_samples) *psnr_samples = samples;
  return dist;
}

/* ========================================================================= *
 *  Interpolation                                                            *
 * ========================================================================= */

/* 1/16-pel luma taps (kLumaFilterHighPrec, inter_prediction.cc:55-73) and
 * 1/32-pel chroma taps (kChromaFilterHighPrec, :91-126): the normative
 * interpolation filters of the format. */
static const int16_t xo_luma_taps[16][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0},       {0, 1, -3, 63, 4, -2, 1, 0},
    {-1, 2, -5, 62, 8, -3, 1, 0},    {-1, 3, -8, 60, 13, -4, 1, 0},
    {-1, 4, -10, 58, 17, -5, 1, 0},  {-1, 4, -11, 52, 26, -8, 3, -1},
    {-1, 3, -9, 47, 31, -10, 4, -1}, {-1, 4, -11, 45, 34, -10, 4, -1},
    {-1, 4, -11, 40, 40, -11, 4, -1}, {-1, 4, -10, 34, 45, -11, 4, -1},
    {-1, 4, -10, 31, 47, -9, 3, -1}, {-1, 3, -8, 26, 52, -11, 4, -1},
    {0, 1, -5, 17, 58, -10, 4, -1},  {0, 1, -4, 13, 60, -8, 3, -1},
    {0, 1, -3, 8, 62, -5, 2, -1},    {0, 1, -2, 4, 63, -3, 1, 0}};

static const int16_t xo_chroma_taps[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},
    {-2, 58, 10, -2}, {-3, 57, 12, -2}, {-4, 56, 14, -2}, {-4, 55, 15, -2},
    {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3},
    {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4},
    {-4, 36, 36, -4}, {-4, 33, 39, -4}, {-4, 30, 42, -4}, {-4, 29, 44, -5},
    {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3},
    {-2, 10, 58, -2}, {-1, 7, 60, -2},  {0, 4, 62, -2},   {0, 2, 63, -1}};

enum { XO_PREC = 14, XO_FPREC = 6, XO_IOFF = 1 << 13 }; /* inter_prediction.h:60-62 */

/* Generic 1-D FIR over `taps` taps with element step `step` (1 = horizontal,
 * stride = vertical).  src_kind: 0 = Sample in, 1 = int16 in.
 * dst_kind: 0 = Sample out (clip), 1 = int16 out.  Shift/offset rules:
 * inter_prediction.h:218-254; loops: inter_prediction.cc:1207-1385.
 * Note :1364 passes `bitdepth` to GetFilterOffset<int16_t,false> which returns
 * 0 regardless. */
static void xo_fir(int taps, int src_short, int dst_short, int bd, int w, int h,
                   const int16_t *f, const void *src_v, ptrdiff_t ss,
                   ptrdiff_t step, void *dst_v, ptrdiff_t ds) {
  int shift, offset;
  const int head = XO_PREC - bd;
  if (!src_short && !dst_short) { /* Sample -> Sample */
    shift = XO_FPREC;
    offset = 1 << (shift - 1);
  } else if (!src_short && dst_short) { /* Sample -> int16 */
    shift = XO_FPREC - head;
    offset = -(XO_IOFF << shift);
  } else if (src_short && !dst_short) { /* int16 -> Sample */
    shift = XO_FPREC + head;
    offset = (XO_IOFF << XO_FPREC) + (1 << (shift - 1));
  } else { /* int16 -> int16 */
    shift = XO_FPREC;
    offset = 0;
  }
  const int smax = (1 << bd) - 1;
  const uint16_t *s16 = (const uint16_t *)src_v;
  const int16_t *i16 = (const int16_t *)src_v;
  const ptrdiff_t back = (taps / 2 - 1) * step;
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      int sum = 0;
      for (int k = 0; k < taps; k++) {
        ptrdiff_t idx = y * ss + x - back + k * step;
        int v = src_short ? (int)i16[idx] : (int)s16[idx];
        sum += v * f[k];
      }
      int val = (sum + offset) >> shift;
      if (dst_short) {
        ((int16_t *)dst_v)[y * ds + x] = (int16_t)val;
      } else if (step != 1) {
        /* vertical -> Sample paths narrow to int16 before the clip
         * (inter_prediction.cc:1290, :1346) */
        ((uint16_t *)dst_v)[y * ds + x] = xo_clip_bd((int16_t)val, smax);
      } else {
        ((uint16_t *)dst_v)[y * ds + x] = xo_clip_bd(val, smax);
      }
    }
  }
}

/* FilterLuma / FilterChroma (+Bipred), inter_prediction.cc:1387-1538 */
static void xo_filter_2d(int bd, int is_chroma, int dst_short, int w, int h,
                         int frac_x, int frac_y, const uint16_t *ref,
                         ptrdiff_t rs, void *pred, ptrdiff_t ps) {
  const int N = is_chroma ? 4 : 8;
  const int16_t *fh = is_chroma ? xo_chroma_taps[frac_x] : xo_luma_taps[frac_x];
  const int16_t *fv = is_chroma ? xo_chroma_taps[frac_y] : xo_luma_taps[frac_y];
  if (frac_y == 0) {
    xo_fir(N, 0, dst_short, bd, w, h, fh, ref, rs, 1, pred, ps);
  } else if (frac_x == 0) {
    xo_fir(N, 0, dst_short, bd, w, h, fv, ref, rs, rs, pred, ps);
  } else {
    int16_t tmp[XO_MAX_BLK * (XO_MAX_BLK + 7)];
    xo_fir(N, 0, 1, bd, w, h + N - 1, fh, ref - (N / 2 - 1) * rs, rs, 1, tmp,
           w);
    xo_fir(N, 1, dst_short, bd, w, h, fv, tmp + (N / 2 - 1) * w, w, w, pred,
           ps);
  }
}

void xo_mc_uni(int bd, int is_chroma, int w, int h, int frac_x, int frac_y,
               const uint16_t *ref, ptrdiff_t rs, uint16_t *pred,
               ptrdiff_t ps) {
  if (frac_x == 0 && frac_y == 0) { /* CopyFrom, inter_prediction.cc:1144 */
    for (int y = 0; y < h; y++)
      memcpy(pred + y * ps, ref + y * rs, (size_t)w * sizeof(uint16_t));
    return;
  }
  xo_filter_2d(bd, is_chroma, 0, w, h, frac_x, frac_y, ref, rs, pred, ps);
}

void xo_mc_uni_bipred(int bd, int is_chroma, int w, int h, int frac_x,
                      int frac_y, const uint16_t *ref, ptrdiff_t rs,
                      int16_t *pred, ptrdiff_t ps) {
  if (frac_x == 0 && frac_y == 0) {
    /* FilterCopyBipred_c, inter_prediction.cc:1462-1473 */
    const int shift = XO_PREC - bd;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int16_t val = (int16_t)(ref[y * rs + x] << shift);
        pred[y * ps + x] = (int16_t)(val - (int16_t)XO_IOFF);
      }
    return;
  }
  xo_filter_2d(bd, is_chroma, 1, w, h, frac_x, frac_y, ref, rs, pred, ps);
}

void xo_add_avg(int bd, int w, int h, const int16_t *s1, ptrdiff_t st1,
                const int16_t *s2, ptrdiff_t st2, uint16_t *dst,
                ptrdiff_t ds) {
  /* inter_prediction.cc:1545-1547 */
  const int head = XO_PREC - bd;
  const int shift = (head > 2 ? head : 2) + 1;
  const int offset = (1 << (shift - 1)) + 2 * XO_IOFF;
  const int smax = (1 << bd) - 1;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      dst[y * ds + x] = (uint16_t)xo_clip3(
          (s1[y * st1 + x] + s2[y * st2 + x] + offset) >> shift, 0, smax);
}

void xo_clip_mv(int pos_x, int pos_y, int pic_w, int pic_h, int *mv_x,
                int *mv_y) {
  /* inter_prediction.cc:769-782 */
  const int offset = 8, sh = 4, maxblk = 64;
  int min_x = -((maxblk + offset + pos_x - 1) << sh);
  int min_y = -((maxblk + offset + pos_y - 1) << sh);
  int max_x = (pic_w + offset - pos_x - 1) << sh;
  int max_y = (pic_h + offset - pos_y - 1) << sh;
  *mv_x = xo_clip3(*mv_x, min_x, max_x);
  *mv_y = xo_clip3(*mv_y, min_y, max_y);
}

void xo_mc_block(int bd, int comp, int x, int y, int w, int h, int mv_x,
                 int mv_y, int pic_w, int pic_h, const uint16_t *ref_plane,
                 ptrdiff_t rs, uint16_t *pred, ptrdiff_t ps) {
  /* MotionCompensationMv :740-758, GetFullpelRef :1174-1205 (4:2:0) */
  xo_clip_mv(x, y, pic_w, pic_h, &mv_x, &mv_y);
  const int cs = comp ? 1 : 0; /* chroma shift */
  const int shift = 4 + cs;
  int pel_x = mv_x >> shift, pel_y = mv_y >> shift;
  int frac_x = mv_x & ((1 << shift) - 1);
  int frac_y = mv_y & ((1 << shift) - 1);
  if (comp) { /* << (1 - size_shift) = << 0 for 4:2:0 */
    frac_x <<= (1 - cs);
    frac_y <<= (1 - cs);
  }
  const int cx = x >> cs, cy = y >> cs, cw = w >> cs, ch = h >> cs;
  const uint16_t *ref = ref_plane + (ptrdiff_t)(cy + pel_y) * rs + cx + pel_x;
  xo_mc_uni(bd, comp != 0, cw, ch, frac_x, frac_y, ref, rs, pred, ps);
}

static int xo_affine_subblock(int ref_x, int ref_y, int mv_x, int mv_y, int size,
                              int scale) {
  /* get_subblock_size lambda, inter_prediction.cc:1071-1086 */
  const int dx = abs(mv_x - ref_x), dy = abs(mv_y - ref_y);
  const int max_len = dx > dy ? dx : dy;
  if (!max_len) return size;
  int sb = (size >> 2) / max_len;
  if (sb < 1) sb = 1;
  while (size % sb) sb--;
  return (sb > 4 ? sb : 4) >> scale;
}

void xo_mc_affine_block(int bd, int comp, int x, int y, int w, int h,
                        const int mv_in[3][2], int pic_w, int pic_h,
                        const uint16_t *ref_plane, ptrdiff_t rs, uint16_t *pred,
                        ptrdiff_t ps) {
  /* InterPrediction::MotionCompAffine -> Sample (inter_prediction.cc:1044-1136);
   * x,y,w,h: luma position / size of the CU; 4:2:0 */
  int mv[3][2];
  for (int i = 0; i < 3; i++) { /* ClipMv(MotionVector3), :784-799 */
    mv[i][0] = mv_in[i][0];
    mv[i][1] = mv_in[i][1];
    xo_clip_mv(x, y, pic_w, pic_h, &mv[i][0], &mv[i][1]);
  }
  const int cs = comp ? 1 : 0, shift = 4 + cs;
  const int cx = x >> cs, cy = y >> cs, cw = w >> cs, ch = h >> cs;
  if (mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) { /* :1063-1069 */
    const uint16_t *r = ref_plane + (ptrdiff_t)(cy + (mv[0][1] >> shift)) * rs + cx +
                        (mv[0][0] >> shift);
    xo_mc_uni(bd, comp != 0, cw, ch, mv[0][0] & ((1 << shift) - 1),
              mv[0][1] & ((1 << shift) - 1), r, rs, pred, ps);
    return;
  }
  /* note: sizes passed are the COMPONENT's width / height (:1087-1090) */
  const int sbw = xo_affine_subblock(mv[0][0], mv[0][1], mv[1][0], mv[1][1], cw, cs);
  const int sbh = xo_affine_subblock(mv[0][0], mv[0][1], mv[2][0], mv[2][1], ch, cs);
  const int mv_max_x = (pic_w - x + 8 - 1) * 16, mv_min_x = (-64 - x - 8 + 1) * 16;
  const int mv_max_y = (pic_h - y + 8 - 1) * 16, mv_min_y = (-64 - y - 8 + 1) * 16;
  const int dhx = ((mv[1][0] - mv[0][0]) * 256) / cw; /* C division, :1101-1102 */
  const int dhy = ((mv[1][1] - mv[0][1]) * 256) / cw;
  const int dvx = -dhy, dvy = dhx;
  int hor_x = mv[0][0] * 256, hor_y = mv[0][1] * 256;
  int ver_x = hor_x, ver_y = hor_y;
  for (int sy = 0; sy < ch; sy += sbh) {
    for (int sx = 0; sx < cw; sx += sbw) {
      int mx = (hor_x + dhx * (sbw >> 1) + dvx * (sbh >> 1)) >> 8;
      int my = (hor_y + dhy * (sbw >> 1) + dvy * (sbh >> 1)) >> 8;
      mx = xo_clip3(mx, mv_min_x, mv_max_x);
      my = xo_clip3(my, mv_min_y, mv_max_y);
      const uint16_t *r = ref_plane + (ptrdiff_t)(cy + sy + (my >> shift)) * rs + cx + sx +
                          (mx >> shift);
      xo_mc_uni(bd, comp != 0, sbw, sbh, mx & ((1 << shift) - 1),
                my & ((1 << shift) - 1), r, rs, pred + (ptrdiff_t)sy * ps + sx, ps);
      hor_x += dhx * sbw;
      hor_y += dhy * sbw;
    }
    ver_x += dvx * sbh;
    ver_y += dvy * sbh;
    hor_x = ver_x;
    hor_y = ver_y;
  }
}

/* ========================================================================= *
 *  Transforms                                                               *
 * ========================================================================= */

/* High-precision (8-bit fraction) matrices.  The reference ships them as
 * literal tables (transform_data.cc:109-796); they are the JEM definitions
 *   coef = (int)(256*sqrt(N) * v + (v > 0 ? 0.5 : -0.5))
 * with v the orthonormal DCT-2/5/8, DST-1/7 basis.  tests/ checks every entry
 * against the reference tables (oracle/_ref) and a committed checksum. */
static int16_t *xo_tx_tab[6][7]; /* [type][log2 size] */
static int xo_tx_init_done;

static void xo_tx_init(void) {
  if (xo_tx_init_done) return;
  const double pi = 3.14159265358979323846;
  for (int t = XVC_TX_DCT2; t <= XVC_TX_DST7; t++) {
    for (int l = 1; l <= 6; l++) {
      const int N = 1 << l;
      if (N == 2 && t != XVC_TX_DCT2) continue;
      int16_t *m = (int16_t *)malloc(sizeof(int16_t) * N * N);
      const double s = sqrt((double)N) * 256.0;
      for (int k = 0; k < N; k++) {
        for (int n = 0; n < N; n++) {
          double v;
          double w0 = (k == 0) ? sqrt(0.5) : 1.0;
          double w1 = (n == 0) ? sqrt(0.5) : 1.0;
          switch (t) {
            case XVC_TX_DCT2:
              v = cos(pi * (n + 0.5) * k / N) * w0 * sqrt(2.0 / N);
              break;
            case XVC_TX_DCT5:
              v = cos(pi * n * k / (N - 0.5)) * w0 * w1 * sqrt(2.0 / (N - 0.5));
              break;
            case XVC_TX_DCT8:
              v = cos(pi * (k + 0.5) * (n + 0.5) / (N + 0.5)) *
                  sqrt(2.0 / (N + 0.5));
              break;
            case XVC_TX_DST1:
              v = sin(pi * (n + 1) * (k + 1) / (N + 1)) * sqrt(2.0 / (N + 1));
              break;
            default: /* DST7 */
              v = sin(pi * (k + 0.5) * (n + 1) / (N + 0.5)) *
                  sqrt(2.0 / (N + 0.5));
              break;
          }
          m[k * N + n] = (int16_t)(int)(s * v + (v > 0 ? 0.5 : -0.5));
        }
      }
      xo_tx_tab[t][l] = m;
    }
  }
  xo_tx_init_done = 1;
}

/* XVC_TX_DCT2_LOW: TransformData::kDct2Transform4 .. 32 (transform_data.cc:26-107),
 * the 6-bit DCT-2 of HEVC - what kDefault / kDct2 select for those sizes under
 * Restrictions::disable_ext2_transform_high_precision (transform.cc:458-605).
 * The 32-point basis has 33 magnitudes, xo_dct2_low[j] ~ 64 sqrt(2) cos(j pi / 64)
 * as the standard rounds them; smaller sizes take every (32 / N)-th row. */
static const int16_t xo_dct2_low[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80,
                                        78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46,
                                        43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static int16_t *xo_tx_low[7];
static const int16_t *xo_low_matrix(int l) {
  if (l < 2 || l > 5) return NULL;
  if (!xo_tx_low[l]) {
    const int N = 1 << l;
    int16_t *m = (int16_t *)malloc(sizeof(int16_t) * N * N);
    for (int k = 0; k < N; k++)
      for (int n = 0; n < N; n++) {
        const int a = (k * (2 * n + 1) * (32 / N)) & 127; /* units of pi / 64 */
        m[k * N + n] = (int16_t)(a <= 32 ? xo_dct2_low[a] : a <= 64 ? -xo_dct2_low[64 - a]
                                         : a <= 96 ? -xo_dct2_low[a - 64] : xo_dct2_low[128 - a]);
      }
    xo_tx_low[l] = m;
  }
  return xo_tx_low[l];
}

const int16_t *xo_transform_matrix(int tx_type, int size) {
  xo_tx_init();
  if (tx_type == XVC_TX_DEFAULT) tx_type = XVC_TX_DCT2;
  int l = xo_log2_size(size);
  if ((1 << l) != size || l > 6) return NULL;
  if (tx_type == XVC_TX_DCT2_LOW) return xo_low_matrix(l);
  if (tx_type < 1 || tx_type > 5) return NULL;
  return xo_tx_tab[tx_type][l];
}

/* kTransformHighPrecisionShift of a 1-D stage (transform.cc:91-99, :876-884):
 * absent only for the low-precision DCT-2 */
static int xo_hp_shift(int tx_type) { return tx_type == XVC_TX_DCT2_LOW ? 0 : 2; }
static int xo_is_dct2(int t) {
  return t == XVC_TX_DEFAULT || t == XVC_TX_DCT2 || t == XVC_TX_DCT2_LOW;
}

/* One forward 1-D pass = transposing matrix product (SURVEY appendix C;
 * FwdGenericTransformN transform.cc:1580-1612, partial butterflies
 * :1186-1578 are exact regroupings): reads `lines` rows of N inputs, writes
 * out[k][y] = (sum_j M[k][j]*in[y][j] + add) >> shift, unclipped int16.
 * 64-point keeps outputs 0..31 only; zero_out stage handles min(lines,32). */
static void xo_fwd_1d(const int16_t *M, int N, int shift, int lines,
                      int zero_out, const int16_t *in, ptrdiff_t is,
                      int16_t *out, ptrdiff_t os) {
  const int add = 1 << (shift - 1);
  const int tx_lines = zero_out ? (lines < 32 ? lines : 32) : lines;
  const int out_rows = N < 32 ? N : 32;
  for (int y = 0; y < tx_lines; y++) {
    for (int k = 0; k < out_rows; k++) {
      int32_t sum = 0;
      for (int j = 0; j < N; j++)
        sum = (int32_t)((uint32_t)sum +
                        (uint32_t)((int32_t)M[k * N + j] * in[y * is + j]));
      out[k * os + y] = (int16_t)((int32_t)((uint32_t)sum + (uint32_t)add) >> shift);
    }
  }
  for (int k = 0; k < out_rows; k++)
    for (int y = tx_lines; y < lines; y++) out[k * os + y] = 0;
  for (int k = out_rows; k < N; k++)
    for (int y = 0; y < lines; y++) out[k * os + y] = 0;
}

/* Inverse 1-D pass (InvGenericTransformN transform.cc:835-862): reads input
 * columns, out[y][k] = clip16((sum_j M[j][k]*in[j][y] + add) >> shift). */
static void xo_inv_1d(const int16_t *M, int N, int shift, int lines,
                      int zero_out, const int16_t *in, ptrdiff_t is,
                      int16_t *out, ptrdiff_t os) {
  const int add = 1 << (shift - 1);
  const int tx_lines = zero_out ? (lines < 32 ? lines : 32) : lines;
  const int in_rows = N < 32 ? N : 32;
  for (int y = 0; y < tx_lines; y++) {
    for (int k = 0; k < N; k++) {
      int32_t sum = 0;
      for (int j = 0; j < in_rows; j++)
        sum = (int32_t)((uint32_t)sum +
                        (uint32_t)((int32_t)M[j * N + k] * in[j * is + y]));
      out[y * os + k] = (int16_t)xo_clip3(
          (int32_t)((uint32_t)sum + (uint32_t)add) >> shift, -32768, 32767);
    }
  }
  for (int y = tx_lines; y < lines; y++)
    for (int k = 0; k < N; k++) out[y * os + k] = 0;
}

/* FwdPartialDst4 (transform.cc:997-1017); shift already reduced by 2. */
static void xo_fwd_dst4(int shift, const int16_t *in, ptrdiff_t is,
                        int16_t *out, ptrdiff_t os) {
  const int add = 1 << (shift - 1);
  for (int i = 0; i < 4; i++) {
    int c0 = in[0] + in[3], c1 = in[1] + in[3], c2 = in[0] - in[1];
    int c3 = 74 * in[2];
    out[0 * os] = (int16_t)((29 * c0 + 55 * c1 + c3 + add) >> shift);
    out[1 * os] = (int16_t)((74 * (in[0] + in[1] - in[3]) + add) >> shift);
    out[2 * os] = (int16_t)((29 * c2 + 55 * c0 - c3 + add) >> shift);
    out[3 * os] = (int16_t)((55 * c2 - 29 * c1 + c3 + add) >> shift);
    in += is;
    out++;
  }
}

/* InvPartialDst4 (transform.cc:217-242). */
static void xo_inv_dst4(int shift, const int16_t *in, ptrdiff_t is,
                        int16_t *out, ptrdiff_t os) {
  const int add = 1 << (shift - 1);
  for (int i = 0; i < 4; i++) {
    int c0 = in[0] + in[2 * is], c1 = in[2 * is] + in[3 * is];
    int c2 = in[0] - in[3 * is], c3 = 74 * in[1 * is];
    out[0] = (int16_t)xo_clip3((29 * c0 + 55 * c1 + c3 + add) >> shift, -32768, 32767);
    out[1] = (int16_t)xo_clip3((55 * c2 - 29 * c1 + c3 + add) >> shift, -32768, 32767);
    out[2] = (int16_t)xo_clip3(
        (74 * (in[0] - in[2 * is] + in[3 * is]) + add) >> shift, -32768, 32767);
    out[3] = (int16_t)xo_clip3((55 * c0 + 29 * c2 - c3 + add) >> shift, -32768, 32767);
    in++;
    out += os;
  }
}

void xo_fwd_transform(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                      const int16_t *resi, ptrdiff_t rs, int16_t *coeff,
                      ptrdiff_t cs) {
  /* transform.cc:869-961 */
  int16_t tmp[XO_MAX_BLK * XO_MAX_BLK];
  const int shift1 = xo_log2_size(w) + bd - 9 + xo_hp_shift(tx_hor);
  const int shift2 = xo_log2_size(h) + 6 + xo_hp_shift(tx_ver);
  if (dst4x4 && w == 4 && h == 4) { /* no high precision for the 4x4 DST (:997-1000) */
    xo_fwd_dst4(xo_log2_size(w) + bd - 9, resi, rs, tmp, XO_MAX_BLK);
    xo_fwd_dst4(xo_log2_size(h) + 6, tmp, XO_MAX_BLK, coeff, cs);
    return;
  }
  xo_fwd_1d(xo_transform_matrix(tx_hor, w), w, shift1, h, 0, resi, rs, tmp,
            XO_MAX_BLK);
  xo_fwd_1d(xo_transform_matrix(tx_ver, h), h, shift2, w, 1, tmp, XO_MAX_BLK,
            coeff, cs);
}

void xo_inv_transform(int bd, int w, int h, int tx_hor, int tx_ver, int dst4x4,
                      int dc_only, const int16_t *coeff, ptrdiff_t cs,
                      int16_t *resi, ptrdiff_t rs) {
  /* transform.cc:83-182 */
  int16_t tmp[XO_MAX_BLK * XO_MAX_BLK];
  const int shift1 = 7 + xo_hp_shift(tx_ver);
  const int shift2 = 20 - bd + xo_hp_shift(tx_hor);
  if (dst4x4 && w == 4 && h == 4) {
    xo_inv_dst4(7, coeff, cs, tmp, XO_MAX_BLK);
    xo_inv_dst4(20 - bd, tmp, XO_MAX_BLK, resi, rs);
    return;
  }
  if (dc_only && xo_is_dct2(tx_ver) && xo_is_dct2(tx_hor)) {
    /* InvDct2Dc, transform.cc:279-291 */
    const int shift = 14 - bd;
    const int add = 1 << (shift - 1);
    int16_t c = (int16_t)((((coeff[0] + 1) >> 1) + add) >> shift);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) resi[y * rs + x] = c;
    return;
  }
  /* vertical first (type index 0, over height, zero_out), then horizontal */
  xo_inv_1d(xo_transform_matrix(tx_ver, h), h, shift1, w, 1, coeff, cs, tmp,
            XO_MAX_BLK);
  xo_inv_1d(xo_transform_matrix(tx_hor, w), w, shift2, h, 0, tmp, XO_MAX_BLK,
            resi, rs);
}

static int xo_transform_shift(int w, int h, int bd) { /* quantize.cc:127-131 */
  return 15 - bd - ((xo_log2_size(w) + xo_log2_size(h)) >> 1);
}

void xo_fwd_transform_skip(int bd, int w, int h, const int16_t *resi,
                           ptrdiff_t rs, int16_t *coeff, ptrdiff_t cs) {
  /* transform.cc:963-995 */
  const int bias = (xo_log2_size(w) + xo_log2_size(h)) % 2 != 0;
  const int shift = xo_transform_shift(w, h, bd) + (bias ? -8 : 0);
  const int scale = bias ? 181 : 1;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (shift > 0)
        coeff[y * cs + x] = (int16_t)((resi[y * rs + x] * scale) * (1 << shift));
      else
        coeff[y * cs + x] = (int16_t)(
            (resi[y * rs + x] * scale + (1 << (-shift - 1))) >> -shift);
    }
}

void xo_inv_transform_skip(int bd, int w, int h, const int16_t *coeff,
                           ptrdiff_t cs, int16_t *resi, ptrdiff_t rs) {
  /* transform.cc:184-215 */
  const int bias = (xo_log2_size(w) + xo_log2_size(h)) % 2 != 0;
  const int shift = xo_transform_shift(w, h, bd) + (bias ? 7 : 0);
  const int scale = bias ? 181 : 1;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (shift > 0)
        resi[y * rs + x] =
            (int16_t)((coeff[y * cs + x] * scale + (1 << (shift - 1))) >> shift);
      else
        resi[y * rs + x] = (int16_t)((uint32_t)(coeff[y * cs + x] * scale) << -shift);
    }
}

/* Qp scale tables, quantize.cc:40-46 */
static const int xo_fwd_scales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
static const int xo_inv_scales[6] = {40, 45, 51, 57, 64, 72};

static int xo_qp_bitdepth(int qp_raw, int bd) { /* quantize.cc:58-63 */
  int q = qp_raw + 6 * (bd - 8);
  return q > 0 ? q : 0;
}

void xo_dequant(int bd, int qp_raw, int w, int h, const int16_t *in,
                ptrdiff_t is, int16_t *out, ptrdiff_t os) {
  /* quantize.cc:94-125 */
  const int qpb = xo_qp_bitdepth(qp_raw, bd);
  const int bias = (xo_log2_size(w) + xo_log2_size(h)) % 2 != 0;
  const int shift = 6 - xo_transform_shift(w, h, bd) + (bias ? 8 : 0);
  const int scale = (xo_inv_scales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t prod = (int32_t)((uint32_t)(int32_t)in[y * is + x] * (uint32_t)scale);
      int32_t c;
      if (shift > 0)
        c = (int32_t)((uint32_t)prod + (1u << (shift - 1))) >> shift;
      else
        c = (int32_t)((uint32_t)prod << -shift);
      out[y * os + x] = (int16_t)xo_clip3(c, -32768, 32767);
    }
}

/* Coefficient scan of a 4x4 sub-block (position = y*4 + x) and of the grid of
 * sub-blocks: diagonal = anti-diagonals walked from bottom-left to top-right
 * (TransformHelper::kScanCoeff4x4 / DeriveSubblockScan, transform.cc:72-76,
 * :1639-1683); horizontal = raster, vertical = column-major. */
static void xo_scan_order(int order, int gw, int gh, uint16_t *tab) {
  int n = 0;
  if (order == 0) {
    for (int s = 0; s < gw + gh - 1; s++)
      for (int y = (s < gh ? s : gh - 1); y >= 0 && s - y < gw; y--)
        tab[n++] = (uint16_t)(y * gw + (s - y));
  } else if (order == 1) {
    for (int y = 0; y < gh; y++)
      for (int x = 0; x < gw; x++) tab[n++] = (uint16_t)(y * gw + x);
  } else {
    for (int x = 0; x < gw; x++)
      for (int y = 0; y < gh; y++) tab[n++] = (uint16_t)(y * gw + x);
  }
}

/* RdoQuant::CoeffSignHideFast (rdo_quant.cc:448-573): per 4x4 sub-block, when
 * the first and last non-zero levels are more than 3 scan positions apart,
 * the parity of the level sum must equal the sign bit of the first non-zero
 * level; otherwise the level whose rounding was cheapest to flip moves by 1. */
static int xo_sign_hide_fast(int scan_order, int w, int h, const int16_t *in,
                             ptrdiff_t is, const int16_t *delta, ptrdiff_t ds,
                             int16_t *out, ptrdiff_t os) {
  uint16_t c4[16], sb[256];
  xo_scan_order(scan_order, 4, 4, c4);
  const int gw = w >> 2, gh = h >> 2;
  xo_scan_order(scan_order, gw, gh, sb);
  int nnz = 0, last_subblock = -1;
  for (int i = gw * gh - 1; i >= 0; i--) {
    const int px = (sb[i] % gw) << 2, py = (sb[i] / gw) << 2;
#define AT(buf, stride, k) (buf)[(py + (c4[k] >> 2)) * (stride) + px + (c4[k] & 3)]
    int last = -1, first = 16, sum = 0;
    for (int k = 0; k < 16; k++) {
      const int c = AT(out, os, k);
      if (c) {
        if (k < first) first = k;
        if (k > last) last = k;
        sum += c;
        nnz++;
      }
    }
    if (last >= 0 && last_subblock == -1) last_subblock = 1;
    if (last - first > 3) {
      const int sign = AT(out, os, first) > 0 ? 0 : 1;
      if (sign != (sum & 1)) {
        int16_t curr_cost = 32767, curr_change = 0, min_cost = 32767, min_change = 0;
        int min_index = -1;
        for (int k = (last_subblock == 1) ? last : 15; k >= 0; k--) {
          if (AT(out, os, k) != 0) {
            if (AT(delta, ds, k) > 0) {
              curr_cost = (int16_t)-AT(delta, ds, k);
              curr_change = 1;
            } else if (k == first && abs(AT(out, os, k)) == 1) {
              curr_cost = 32767;
            } else {
              curr_cost = AT(delta, ds, k);
              curr_change = -1;
            }
          } else if (k < first && (AT(in, is, k) >= 0 ? 0 : 1) != sign) {
            curr_cost = 32767;
          } else {
            curr_cost = (int16_t)-AT(delta, ds, k);
            curr_change = 1;
          }
          if (curr_cost < min_cost) {
            min_cost = curr_cost;
            min_change = curr_change;
            min_index = k;
          }
        }
        if (AT(out, os, min_index) == -32768 || AT(out, os, min_index) == 32767)
          min_change = -1;
        if (!AT(out, os, min_index)) nnz++;
        if (AT(in, is, min_index) >= 0)
          AT(out, os, min_index) = (int16_t)(AT(out, os, min_index) + min_change);
        else
          AT(out, os, min_index) = (int16_t)(AT(out, os, min_index) - min_change);
        if (!AT(out, os, min_index)) nnz--;
      }
    }
#undef AT
    if (last_subblock == 1) last_subblock = 0;
  }
  return nnz;
}

int xo_quant_fast2(int bd, int qp_raw, int intra_pic, int sign_hide, int scan_order,
                   int w, int h, const int16_t *in, ptrdiff_t is, int16_t *out,
                   ptrdiff_t os) {
  /* RdoQuant::QuantFast, rdo_quant.cc:156-201 */
  const int qpb = xo_qp_bitdepth(qp_raw, bd);
  const int bias = (xo_log2_size(w) + xo_log2_size(h)) % 2 != 0;
  const int shift = 14 + qpb / 6 + xo_transform_shift(w, h, bd) + (bias ? 7 : 0);
  const int scale = xo_fwd_scales[qpb % 6] * (bias ? 181 : 1);
  const int64_t offset = (int64_t)((intra_pic ? 171ull : 85ull) << (shift - 9));
  static __thread int16_t delta[64 * 64];
  int nnz = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = in[y * is + x];
      int sign = v < 0 ? -1 : 1;
      int64_t abs_coeff = abs(v);
      int level = (int)(((abs_coeff * scale) + offset) >> shift);
      nnz += level != 0;
      out[y * os + x] = (int16_t)xo_clip3(level * sign, -32768, 32767);
      delta[y * 64 + x] =
          (int16_t)(((abs_coeff * scale) - ((int64_t)level << shift)) >> (shift - 8));
    }
  if (sign_hide && nnz > 1 && w >= 4 && h >= 4)
    nnz = xo_sign_hide_fast(scan_order, w, h, in, is, delta, 64, out, os);
  return nnz;
}

int xo_quant_fast(int bd, int qp_raw, int intra_pic, int w, int h,
                  const int16_t *in, ptrdiff_t is, int16_t *out,
                  ptrdiff_t os) {
  /* QuantFast with disable_transform_sign_hiding set */
  return xo_quant_fast2(bd, qp_raw, intra_pic, 0, 0, w, h, in, is, out, os);
}
