/*
 * xvc_oracle_stats.c -- CPU restatement of the decision-free whole-picture
 * passes around the hot path (SURVEY.md section 8f row N4 and the I/O
 * conversions either side of it): input / output sample conversion, the
 * reconstruction CRC, the AQP variance map and the LIC histogram distance.
 *
 * TEST INFRASTRUCTURE ONLY (see xvc_oracle.h).  Pinned against the
 * reference's Resampler, Checksum, CuEncoder::CalcDeltaQpFromVariance and
 * PictureEncoder::DetermineAllowLic through oracle/ref_harness.cc
 * (tests/test_oracle_vs_ref.py).  Paths cited are relative to
 * /root/reference/src.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"

/* Resampler::CopyFromBytesFast / CopyFromBytesWithPadding
 * (xvc_common_lib/resample.cc:152-197, :216-262): bytes of one plane at the
 * input bit depth -> samples at the internal depth (left shift), the columns
 * and rows beyond the input size filled by repeating the last column / row. */
void xo_import_plane(int in_bitdepth, int out_bitdepth, int in_w, int in_h,
                     int out_w, int out_h, const uint8_t *src,
                     ptrdiff_t src_stride_bytes, uint16_t *dst,
                     ptrdiff_t dst_stride) {
  const int upshift = out_bitdepth - in_bitdepth;
  for (int y = 0; y < in_h; y++) {
    const uint8_t *row = src + y * src_stride_bytes;
    uint16_t *d = dst + y * dst_stride;
    for (int x = 0; x < in_w; x++) {
      unsigned v = in_bitdepth == 8 ? row[x]
                                    : (unsigned)row[2 * x] | ((unsigned)row[2 * x + 1] << 8);
      d[x] = (uint16_t)(v << upshift);
    }
    for (int x = in_w; x < out_w; x++) d[x] = d[in_w - 1];
  }
  for (int y = in_h; y < out_h; y++)
    memcpy(dst + y * dst_stride, dst + (in_h - 1) * dst_stride,
           sizeof(uint16_t) * out_w);
}

/* Resampler::CopyToBytesWithShift (resample.cc:304-338) with the four sample
 * functions it dispatches to (:475-551): copy, up-shift, rounding down-shift
 * and the error-feedback ("dither") down-shift whose remainder runs on across
 * the rows of the plane.  Output is tightly packed (stride = w), 1 byte per
 * sample when out_bitdepth <= 8, else 2 (little endian). */
void xo_export_plane(int src_bitdepth, int out_bitdepth, int dither, int w, int h,
                     const uint16_t *src, ptrdiff_t src_stride, uint8_t *out) {
  const int wide = out_bitdepth > 8;
  const int smax = (1 << out_bitdepth) - 1;
  int carry = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = src[y * src_stride + x], o;
      if (out_bitdepth >= src_bitdepth || (!wide && src_bitdepth <= 8)) {
        o = wide ? (uint16_t)(v << (out_bitdepth - src_bitdepth)) : (uint8_t)v;
      } else {
        const int shift = src_bitdepth - out_bitdepth;
        if (dither) {
          carry += v;
          o = carry >> shift;
          carry &= (1 << shift) - 1;
        } else {
          o = (v + (1 << (shift - 1))) >> shift;
        }
        o = o < 0 ? 0 : (o > smax ? smax : o);
      }
      if (wide) {
        out[2 * ((size_t)y * w + x)] = (uint8_t)(o & 0xff);
        out[2 * ((size_t)y * w + x) + 1] = (uint8_t)(o >> 8);
      } else {
        out[(size_t)y * w + x] = (uint8_t)o;
      }
    }
}

/* Checksum::CalculateCrc (xvc_common_lib/checksum.cc:46-92): CRC-16, polynomial
 * 0x1021, register preset 0xffff, message bits MSB first - per sample the low
 * byte, then (bit depth > 8) the high byte - and 16 zero bits pushed through
 * at the end.  mode 0 (kMinOverhead): one value over Y,U,V; mode 1
 * (kMaxRobust): one value per component.  Writes 2 bytes per value (high
 * byte first); returns the number of bytes. */
static uint32_t crc_push(uint32_t crc, unsigned byte) {
  for (int bit = 0; bit < 8; bit++) {
    const uint32_t msb = (crc >> 15) & 1;
    const uint32_t b = (byte >> (7 - bit)) & 1;
    crc = (((crc << 1) + b) & 0xffff) ^ (msb * 0x1021);
  }
  return crc;
}

int xo_picture_crc(int bitdepth, int mode, int w, int h,
                   const uint16_t *const planes[3], const ptrdiff_t strides[3],
                   uint8_t *hash) {
  uint32_t crc = 0xffff;
  int n = 0;
  for (int c = 0; c < 3; c++) {
    const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h;
    if (mode == 1) crc = 0xffff;
    for (int y = 0; y < ch; y++)
      for (int x = 0; x < cw; x++) {
        const unsigned v = planes[c][y * strides[c] + x];
        crc = crc_push(crc, v & 0xff);
        if (bitdepth > 8) crc = crc_push(crc, v >> 8);
      }
    if (mode == 1 || c == 2) {
      crc = crc_push(crc_push(crc, 0), 0);
      hash[n++] = (uint8_t)((crc >> 8) & 0xff);
      hash[n++] = (uint8_t)(crc & 0xff);
    }
  }
  return n;
}

/* calc_variance of CuEncoder::CalcDeltaQpFromVariance
 * (xvc_enc_lib/cu_encoder.cc:319-333) for every 16x16 block whose origin is
 * inside the picture: 256 * (sum(x^2) - sum(x)^2 / 256) / 256 in unsigned
 * 64-bit arithmetic.  A block that hangs over the right / bottom edge reads
 * the samples stored there (the caller's border). out: ceil(h/16) rows of
 * ceil(w/16). */
void xo_variance_map(int w, int h, const uint16_t *luma, ptrdiff_t stride,
                     uint64_t *out) {
  const int bw = (w + 15) / 16, bh = (h + 15) / 16;
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      uint64_t sum = 0, squares = 0;
      const uint64_t num = 256;
      for (int k = 0; k < 16; k++)
        for (int l = 0; l < 16; l++) {
          const uint64_t v = luma[(by * 16 + k) * stride + bx * 16 + l];
          sum += v;
          squares += v * v;
        }
      out[by * bw + bx] = (256 * (squares - (sum * sum) / num)) / num;
    }
}

static int cmp_u64(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* The block statistic of one CTU (cu_encoder.cc:335-357): the in-picture
 * 16x16 variances of the ctu_size x ctu_size area at (x, y), sorted; returns
 * 1 + v[blocks / 2].  var_map as written by xo_variance_map. */
uint64_t xo_ctu_variance(int w, int h, int x, int y, int ctu_size,
                         const uint64_t *var_map) {
  const int bw = (w + 15) / 16, n = ctu_size / 16;
  uint64_t v[64];
  int blocks = 0;
  for (int i = 0; i < n; i++) {
    if (y + i * 16 >= h) continue;
    for (int j = 0; j < n; j++) {
      if (x + j * 16 >= w) continue;
      v[blocks++] = var_map[(y / 16 + i) * bw + x / 16 + j];
    }
  }
  qsort(v, blocks, sizeof(v[0]), cmp_u64);
  return 1 + v[blocks / 2];
}

/* ... and the QP offset derived from it (cu_encoder.cc:308-318, :359-363);
 * floating point, host side in the product. */
int xo_aqp_delta_qp(uint64_t ctu_variance, int bitdepth, int aqp_strength) {
  const double strength = 1.0 * aqp_strength / 10;
  const double dqp =
      strength * (1.5 * log((double)ctu_variance) - 15 - 2 * (bitdepth - 8));
  const int v = (int)dqp;
  return v < -3 ? -3 : (v > 7 ? 7 : v);
}

/* PictureEncoder::DetermineAllowLic (xvc_enc_lib/picture_encoder.cc:230-281):
 * sum over the sample values of |histogram(a) - histogram(b)| of two luma
 * planes; LIC is allowed when it exceeds (int)(0.06 * w * h). */
int64_t xo_histogram_distance(int bitdepth, int w, int h, const uint16_t *a,
                              ptrdiff_t sa, const uint16_t *b, ptrdiff_t sb) {
  const int buckets = 1 << bitdepth;
  int32_t *hist = (int32_t *)calloc(buckets, sizeof(int32_t));
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      hist[a[y * sa + x]]++;
      hist[b[y * sb + x]]--;
    }
  int64_t sum = 0;
  for (int i = 0; i < buckets; i++) sum += hist[i] < 0 ? -hist[i] : hist[i];
  free(hist);
  return sum;
}

int xo_allow_lic(int64_t histogram_distance, int w, int h) {
  return histogram_distance > (int)(0.06 * w * h);
}
