/* xvc_oracle_affine_me.c -- CPU restatement (TEST INFRASTRUCTURE, never linked
 * into the product) of the affine motion estimation of the xvc encoder:
 * InterSearch::MotionEstAffine (xvc_enc_lib/inter_search.cc:664-749) and
 * InterSearch::AffineGradientSearch (:751-851), with InterPrediction::
 * DeriveMvAffine (xvc_common_lib/inter_prediction.cc:615-630).
 *
 * Pinned against the reference build: tests/test_oracle_vs_ref.py drives
 * xr_affine_gradient_search / xr_affine_me of oracle/ref_harness.cc (the
 * reference's own member functions) on the same inputs. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"

/* ::lround as the x86-64 build of the reference evaluates it when the value
 * does not fit (cvtsd2si "integer indefinite", whose low 32 bits are 0): the
 * MvDelta members are int. */
static int lround_to_int(double v) {
  if (!(fabs(v) < 9.2e18)) return 0;
  return (int)lround(v);
}

/* inter_search.cc:751-851.  mvd = {mvd0.x, mvd0.y, mvd1.x, mvd1.y} (1/4 pel). */
void xo_affine_gradient_search(int width, int height, const uint16_t *pred, ptrdiff_t ps,
                               const int16_t *err, ptrdiff_t es, int mvd[4]) {
  static float dh[64][64], dv[64][64];
#pragma omp threadprivate(dh, dv)
  const uint16_t *p = pred + ps;
  for (int y = 1; y < height - 1; y++) {
    for (int x = 1; x < width - 1; x++) {
      int a0 = p[x - ps - 1], a1 = p[x - ps], a2 = p[x - ps + 1];
      int b0 = p[x - 1], b2 = p[x + 1];
      int c0 = p[x + ps - 1], c1 = p[x + ps], c2 = p[x + ps + 1];
      dh[y][x] = (-a0 + a2 - 2 * b0 + 2 * b2 - c0 + c2) / 8.0f;
      dv[y][x] = (-a0 - 2 * a1 - a2 + c0 + 2 * c1 + c2) / 8.0f;
    }
    dh[y][0] = dh[y][1];
    dh[y][width - 1] = dh[y][width - 2];
    dv[y][0] = dv[y][1];
    dv[y][width - 1] = dv[y][width - 2];
    p += ps;
  }
  for (int x = 0; x < width; x++) {
    dh[0][x] = dh[1][x];
    dh[height - 1][x] = dh[height - 2][x];
    dv[0][x] = dv[1][x];
    dv[height - 1][x] = dv[height - 2][x];
  }
  double m[4][5];
  memset(m, 0, sizeof(m));
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      const double c[4] = {
          dh[y][x],
          x * dh[y][x] + y * dv[y][x],
          dv[y][x],
          y * dh[y][x] - x * dv[y][x],
      };
      for (int row = 0; row < 4; row++) {
        for (int col = 0; col < 4; col++) m[row][col] += c[row] * c[col];
        m[row][4] += err[x] * c[row];
      }
    }
    err += es;
  }
  /* row echelon form with partial pivoting */
  for (int i = 0; i < 3; i++) {
    int best = i;
    double best_val = fabs(m[i][i]);
    for (int j = i + 1; j < 4; j++)
      if (fabs(m[j][i]) > best_val) {
        best = j;
        best_val = fabs(m[j][i]);
      }
    if (best != i)
      for (int col = 0; col < 5; col++) {
        double t = m[i][col];
        m[i][col] = m[best][col];
        m[best][col] = t;
      }
    for (int j = i + 1; j < 4; j++)
      for (int k = i + 1; k < 5; k++)
        if (m[i][i]) m[j][k] -= m[i][k] * m[j][i] / m[i][i];
  }
  double params[4] = {0, 0, 0, 0};
  if (m[3][3]) params[3] = m[3][4] / m[3][3];
  for (int row = 2; row >= 0; row--) {
    double sum = 0;
    for (int col = row + 1; col < 4; col++) sum += m[row][col] * params[col];
    if (m[row][row]) params[row] = (m[row][4] - sum) / m[row][row];
  }
  const int scale = 4; /* 1 << MvDelta::kPrecisionShift */
  mvd[0] = lround_to_int(scale * params[0]);
  mvd[1] = lround_to_int(scale * params[2]);
  mvd[2] = lround_to_int(scale * (params[1] * width + params[0]));
  mvd[3] = lround_to_int(scale * (-params[3] * width + params[2]));
}

/* inter_prediction.cc:615-630 */
static void derive_mv_affine(const xvcgpu_affine_me_block *b, int pic_w, int pic_h,
                             int mv[3][2]) {
  xo_clip_mv(b->x, b->y, pic_w, pic_h, &mv[0][0], &mv[0][1]);
  xo_clip_mv(b->x, b->y, pic_w, pic_h, &mv[1][0], &mv[1][1]);
  mv[2][0] = mv[0][0] - (mv[1][1] - mv[0][1]) * b->h / b->w;
  mv[2][1] = mv[0][1] + (mv[1][0] - mv[0][0]) * b->h / b->w;
  xo_clip_mv(b->x, b->y, pic_w, pic_h, &mv[2][0], &mv[2][1]);
}

/* GetMvdBits(MotionVector3, MotionVector3, 0), inter_search.cc:1161-1165 */
static uint32_t mvd_bits3(const int mvp[3][2], const int mv[3][2]) {
  return xo_mvd_bits(mvp[0][0], mvp[0][1], mv[0][0], mv[0][1], 0) +
         xo_mvd_bits(mvp[1][0], mvp[1][1], mv[1][0], mv[1][1], 0);
}

/* inter_search.cc:664-749.  orig / ref / ref_other point at sample (0,0) of
 * the luma planes (references padded); ref_other only with XVC_AFFINE_ME_BIPRED:
 * the target is then 2 * orig - MotionCompAffine(ref_other, other_mv)
 * (SearchBiIterative, :394-435; SubtractWeighted, sample_buffer.h:147-161). */
void xo_affine_me(int bd, const xvcgpu_affine_me_block *b, int pic_w, int pic_h,
                  const uint16_t *orig, ptrdiff_t os, const uint16_t *ref, ptrdiff_t rs,
                  const uint16_t *ref_other, ptrdiff_t ros, xvcgpu_affine_me_result *out) {
  const int w = b->w, h = b->h;
  const uint32_t lambda = b->lambda16;
  const int bipred = (b->flags & XVC_AFFINE_ME_BIPRED) != 0;
  const int bi_shift = bipred ? 1 : 0, max_iterations = bipred ? 5 : 7;
  uint16_t pred[64 * 64];
  int16_t err[64 * 64], target[64 * 64];
  const uint16_t *o = orig + (ptrdiff_t)b->y * os + b->x;
  int mvp[3][2], best_mv[3][2], mv[3][2];
  memcpy(mvp, b->mvp, sizeof(mvp));
  memcpy(best_mv, mvp, sizeof(mvp));
#define MC(v) xo_mc_affine_block(bd, 0, b->x, b->y, w, h, (const int (*)[2])(v), pic_w, pic_h, \
                                 ref, rs, pred, 64)
#define DIST(metric)                                                                 \
  ((bipred ? xo_metric_rs(metric, bd, 0, 0, 1.0, w, h, target, 64, pred, 64)         \
           : xo_metric_ss(metric, bd, 0, 0, 1.0, w, h, o, os, pred, 64)) >> bi_shift)
  if (bipred) {
    xo_mc_affine_block(bd, 0, b->x, b->y, w, h, (const int (*)[2])b->other_mv, pic_w, pic_h,
                       ref_other, ros, pred, 64);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        target[y * 64 + x] = (int16_t)(2 * (int)o[y * os + x] - (int)pred[y * 64 + x]);
  }
  MC(mvp);
  uint64_t best_dist = DIST(XVC_METRIC_SAD);
  uint64_t best_cost = best_dist + ((uint32_t)(lambda * mvd_bits3(mvp, best_mv)) >> 16);
  if ((b->flags & XVC_AFFINE_ME_HAS_BOOTSTRAP) &&
      memcmp(b->bootstrap, best_mv, sizeof(best_mv)) != 0) {
    int boot[3][2];
    memcpy(boot, b->bootstrap, sizeof(boot));
    MC(boot);
    uint64_t dist = DIST(XVC_METRIC_SAD);
    uint64_t cost = dist + ((uint32_t)(lambda * mvd_bits3(mvp, boot)) >> 16);
    if (cost < best_cost || bipred) /* force_mv_bootstrap */
      memcpy(best_mv, boot, sizeof(boot));
    else
      MC(best_mv);
  }
  best_dist = DIST(XVC_METRIC_SATD);
  best_cost = best_dist + ((uint32_t)(lambda * mvd_bits3(mvp, best_mv)) >> 16);
  memcpy(mv, best_mv, sizeof(mv));
  uint32_t iterations = 0;
  for (int iter = 0; iter < max_iterations; iter++) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        err[y * 64 + x] = (int16_t)((bipred ? (int)target[y * 64 + x] : (int)o[y * os + x]) -
                                    (int)pred[y * 64 + x]);
    int mvd[4];
    xo_affine_gradient_search(w, h, pred, 64, err, 64, mvd);
    if (!mvd[0] && !mvd[1] && !mvd[2] && !mvd[3]) break;
    iterations++;
    /* MotionVector += MvDelta: 1/4 pel -> 1/16 pel */
    mv[0][0] += mvd[0] * 4;
    mv[0][1] += mvd[1] * 4;
    mv[1][0] += mvd[2] * 4;
    mv[1][1] += mvd[3] * 4;
    derive_mv_affine(b, pic_w, pic_h, mv);
    MC(mv);
    uint64_t dist = DIST(XVC_METRIC_SATD);
    uint64_t cost = dist + ((uint32_t)(lambda * mvd_bits3(mvp, mv)) >> 16);
    if (cost < best_cost) {
      best_cost = cost;
      best_dist = dist;
      memcpy(best_mv, mv, sizeof(mv));
    }
  }
#undef MC
#undef DIST
  memcpy(out->mv, best_mv, sizeof(best_mv));
  out->dist = (uint32_t)best_dist;
  out->iterations = iterations;
}
