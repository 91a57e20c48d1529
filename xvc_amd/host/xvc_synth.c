/* xvc_synth.c -- the deterministic integer-only synthetic clip generator of
 * xvc_amd/synth.py in plain C (shipped in both languages
 * so a C/C++ host and the Python harness feed identical bytes).  Input
 * generation only: libxvcgpu.so does not link it; the oracle's shared object
 * and the C++ host programs compile it in.
 *
 * frame n = crop of a static textured base plane at (2n mod 64, n mod 64) + one
 * 32x32 inverted-contrast square moving (5,3) px/frame + fresh +-2 noise from a
 * 32-bit LCG; U,V = affine functions of the 2x2-averaged luma; 8-bit content
 * returned at the internal bit depth (<< (bd - 8)). */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

static int64_t floordiv(int64_t a, int64_t b) { /* b > 0 */
  int64_t q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
static int64_t tri(int64_t v, int64_t period) {
  int64_t m = ((v % period) + period) % period;
  return m < period - m ? m : period - m;
}
static int64_t clip255(int64_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* element (r, c) of the per-pixel LCG field: row state seeded from a hashed
 * row index, stepped c + 1 times along the row */
static void lcg_row(uint64_t seed, int r, int cols, uint32_t *out) {
  uint32_t s = (uint32_t)((uint64_t)r * 2654435761ull + seed);
  for (int c = 0; c < cols; c++) {
    s = s * 1664525u + 1013904223u;
    out[c] = s;
  }
}

/* Y: height x width, U/V: (height/2) x (width/2); strides in samples. */
void xo_synth_frame(int width, int height, int bitdepth, uint64_t seed, int square,
                    int n, uint16_t *Y, ptrdiff_t ys, uint16_t *U, ptrdiff_t us,
                    uint16_t *V, ptrdiff_t vs) {
  const int ox = (2 * n) % 64, oy = n % 64;
  const int sx = (40 + 5 * n) % (width - 32 > 1 ? width - 32 : 1);
  const int sy = (24 + 3 * n) % (height - 32 > 1 ? height - 32 : 1);
  const int W = width + 128;
  const int sh = bitdepth - 8;
  uint32_t *base_noise = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)W);
  uint32_t *frame_noise = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)width);
  int64_t *y8 = (int64_t *)malloc(sizeof(int64_t) * (size_t)width * height);
  for (int r = 0; r < height; r++) {
    const int64_t yy = oy + r;
    lcg_row(seed, (int)yy, W, base_noise);
    lcg_row(seed + 7919ull * (uint64_t)(n + 1), r, width, frame_noise);
    for (int c = 0; c < width; c++) {
      const int64_t xx = ox + c;
      const int64_t low = floordiv(tri(xx * 3 + yy, 211) * 120, 105) +
                          floordiv(tri(yy * 5 - xx, 157) * 60, 78);
      const int64_t high = floordiv(tri(xx + 2 * yy, 14) * 30, 7);
      const int64_t noise = (int64_t)(base_noise[xx] >> 27) - 16;
      int64_t v = clip255(30 + low + high + floordiv(noise, 2));
      if (square && r >= sy && r < sy + 32 && c >= sx && c < sx + 32) v = 255 - v;
      const int64_t nz = (int64_t)(frame_noise[c] >> 30);
      v = clip255(v + nz - 2 + (nz == 0));
      y8[(size_t)r * width + c] = v;
      Y[r * ys + c] = (uint16_t)(v << sh);
    }
  }
  for (int r = 0; r < height / 2; r++)
    for (int c = 0; c < width / 2; c++) {
      const int64_t *p = y8 + (size_t)(2 * r) * width + 2 * c;
      const int64_t sub = (p[0] + p[width] + p[1] + p[width + 1] + 2) >> 2;
      U[r * us + c] = (uint16_t)(clip255(128 + floordiv(sub - 128, 3)) << sh);
      V[r * vs + c] = (uint16_t)(clip255(128 - floordiv(sub - 128, 4)) << sh);
    }
  free(base_noise);
  free(frame_noise);
  free(y8);
}
