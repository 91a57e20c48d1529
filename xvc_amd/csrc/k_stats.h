// Whole-picture passes around the hot path (SURVEY.md 8f row N4 and the sample
// conversions either side of it): input / output conversion (Resampler,
// xvc_common_lib/resample.cc:152-262, :304-338, :475-551), reconstruction CRC
// (Checksum::CalculateCrc, checksum.cc:46-92), the AQP variance statistic
// (CuEncoder::CalcDeltaQpFromVariance, xvc_enc_lib/cu_encoder.cc:308-357) and
// the LIC histogram distance (PictureEncoder::DetermineAllowLic,
// picture_encoder.cc:230-281).  All HBM-bound, decision-free, row-shardable.
#ifndef XVCGPU_K_STATS_H_
#define XVCGPU_K_STATS_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

struct __attribute__((packed, aligned(1))) StU8x8 { uint8_t v[8]; };
struct __attribute__((packed, aligned(2))) StU16x8 { uint16_t v[8]; };

// ---- import: packed planar bytes -> samples at the internal depth ----------
// grid: (rows of the destination plane, 3); block 256.  A workgroup writes one
// destination row: source row min(y, in_h - 1), columns beyond in_w repeat the
// last sample (CopyFromBytesWithPadding).  8 samples per thread per sweep.
struct ImportArgs {
  const uint8_t *src[3];  // first byte of each input plane
  int in_w[3], in_h[3];
  int wide;               // input samples are 16-bit little endian
  int upshift;
};

__global__ void __launch_bounds__(256)
picture_import_kernel(PicView dst, ImportArgs a) {
  const int c = blockIdx.y;
  const PlaneView d = dst.c[c];
  const int y = blockIdx.x;
  if (y >= d.h) return;
  const int in_w = a.in_w[c], ys = y < a.in_h[c] ? y : a.in_h[c] - 1;
  const size_t bps = a.wide ? 2 : 1;
  const uint8_t *row = a.src[c] + (size_t)ys * in_w * bps;
  uint16_t *out = d.p + (ptrdiff_t)y * d.stride;
  for (int x0 = threadIdx.x * 8; x0 < d.w; x0 += blockDim.x * 8) {
    uint32_t o[4];  // 8 output samples, packed pairs (kept in registers: no local array
                    // whose address is taken - that one the compiler moves to LDS)
    if (x0 + 8 <= in_w) {
      const uint8_t *sp = row + bps * x0;
      if (a.wide) {
        uint32_t q[4];
        if ((reinterpret_cast<uintptr_t>(sp) & 3) == 0) {  // the usual case: dword loads
          const uint4 t = *reinterpret_cast<const uint4 *>(sp);
          q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
        } else {
          const StU16x8 t = *reinterpret_cast<const StU16x8 *>(sp);
#pragma unroll
          for (int k = 0; k < 4; k++) q[k] = (uint32_t)t.v[2 * k] | ((uint32_t)t.v[2 * k + 1] << 16);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = (((q[k] & 0xffffu) << a.upshift) & 0xffffu) | ((q[k] >> 16) << (16 + a.upshift));
      } else {
        uint32_t q[2];
        if ((reinterpret_cast<uintptr_t>(sp) & 3) == 0) {
          const uint2 t = *reinterpret_cast<const uint2 *>(sp);
          q[0] = t.x; q[1] = t.y;
        } else {
          const StU8x8 t = *reinterpret_cast<const StU8x8 *>(sp);
#pragma unroll
          for (int k = 0; k < 2; k++)
            q[k] = (uint32_t)t.v[4 * k] | ((uint32_t)t.v[4 * k + 1] << 8) |
                   ((uint32_t)t.v[4 * k + 2] << 16) | ((uint32_t)t.v[4 * k + 3] << 24);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t lo = (q[k >> 1] >> (16 * (k & 1))) & 0xffu;
          const uint32_t hi = (q[k >> 1] >> (16 * (k & 1) + 8)) & 0xffu;
          o[k] = (lo << a.upshift) | (hi << (16 + a.upshift));
        }
      }
    } else {  // the chunk that holds the last input column, and the padding
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t pair = 0;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int x = x0 + 2 * k + e < in_w ? x0 + 2 * k + e : in_w - 1;
          const uint32_t sv =
              a.wide ? (uint32_t)row[2 * x] | ((uint32_t)row[2 * x + 1] << 8) : row[x];
          pair |= ((sv << a.upshift) & 0xffffu) << (16 * e);
        }
        o[k] = pair;
      }
    }
    // plane widths are multiples of 4 (picture widths of 8): x0 + 8 may only
    // exceed the row by 4
    if (x0 + 8 <= d.w) {
      *reinterpret_cast<uint4 *>(out + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      *reinterpret_cast<uint2 *>(out + x0) = make_uint2(o[0], o[1]);
    }
  }
}

// ---- export: samples -> packed planar bytes at the output depth -------------
// mode 0: copy / up-shift / plain byte copy, 1: rounding down-shift,
// 2: error-feedback down-shift.  For mode 2 the remainder that reaches sample n
// is (sum of all earlier samples of the plane) mod 2^shift - a prefix sum, so
// it is computed in parallel: row sums (kernel A), their exclusive scan per
// plane (kernel B), and a scan inside the row (kernel C = this kernel).
struct ExportArgs {
  uint8_t *dst[3];  // first byte of each output plane
  int w[3], h[3];   // display size per plane
  int wide;         // output samples are 16-bit
  int mode, shift, smax;
  const uint32_t *row_carry;  // mode 2: remainder entering each row
  int row_base[3];            // index of the plane's first row in row_carry
};

__global__ void __launch_bounds__(256)
export_row_sums_kernel(PicView src, ExportArgs a, uint32_t *row_sum) {
  __shared__ uint32_t part[4];
  const int c = blockIdx.y, y = blockIdx.x;
  if (y >= a.h[c]) return;
  const uint16_t *row = src.c[c].p + (ptrdiff_t)y * src.c[c].stride;
  uint32_t s = 0;
  for (int x = threadIdx.x; x < a.w[c]; x += 256) s += row[x];
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    row_sum[a.row_base[c] + y] = (part[0] + part[1] + part[2] + part[3]) & ((1u << a.shift) - 1);
}

// grid: 3 (planes); block 256: exclusive scan of the row sums, in place.
__global__ void __launch_bounds__(256)
export_row_scan_kernel(ExportArgs a, uint32_t *row_sum) {
  __shared__ uint32_t tot[256];
  const int c = blockIdx.x, n = a.h[c];
  uint32_t *v = row_sum + a.row_base[c];
  const int per = (n + 255) / 256;
  const int b = threadIdx.x * per;
  uint32_t s = 0;
  for (int i = b; i < b + per && i < n; i++) s += v[i];
  tot[threadIdx.x] = s;
  __syncthreads();
  uint32_t before = 0;
  for (int t = 0; t < (int)threadIdx.x; t++) before += tot[t];
  const uint32_t mask = (1u << a.shift) - 1;
  for (int i = b; i < b + per && i < n; i++) {
    const uint32_t x = v[i];
    v[i] = before & mask;
    before += x;
  }
}

__global__ void __launch_bounds__(256)
picture_export_kernel(PicView src, ExportArgs a) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  const int c = blockIdx.y, y = blockIdx.x;
  if (y >= a.h[c]) return;
  const int w = a.w[c];
  const uint16_t *row = src.c[c].p + (ptrdiff_t)y * src.c[c].stride;
  uint8_t *out = a.dst[c] + (size_t)y * w * (a.wide ? 2 : 1);
  const uint32_t mask = (1u << a.shift) - 1;
  if (a.mode == 2 && threadIdx.x == 0) carry_s = a.row_carry[a.row_base[c] + y];
  for (int base = 0; base < w; base += 256 * 8) {
    const int x0 = base + threadIdx.x * 8;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = x0 + k < w ? row[x0 + k] : 0;
    uint32_t o[8];
    if (a.mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = a.wide ? (v[k] << a.shift) & 0xffff : v[k] & 0xff;
    } else if (a.mode == 1) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t r = (v[k] + (1u << (a.shift - 1))) >> a.shift;
        o[k] = r > (uint32_t)a.smax ? a.smax : r;
      }
    } else {
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += v[k];
      // exclusive scan of the per-thread sums over the workgroup
      uint32_t inc = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, XVC_WAVE);
        if ((int)(threadIdx.x & 63) >= d) inc += t;
      }
      __syncthreads();  // carry_s / wave_tot of the previous sweep consumed
      if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
      __syncthreads();
      uint32_t before = carry_s + inc - s;
      for (int q = 0; q < (int)(threadIdx.x >> 6); q++) before += wave_tot[q];
      uint32_t carry = before & mask;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        carry += v[k];
        const uint32_t r = carry >> a.shift;
        o[k] = r > (uint32_t)a.smax ? a.smax : r;
        carry &= mask;
      }
      __syncthreads();
      if (threadIdx.x == 255) carry_s = carry;  // wave_tot rewritten after the next barrier
    }
    if (x0 + 8 <= w) {
      if (a.wide) {
        StU16x8 s;
#pragma unroll
        for (int k = 0; k < 8; k++) s.v[k] = (uint16_t)o[k];
        *reinterpret_cast<StU16x8 *>(out + 2 * x0) = s;
      } else {
        StU8x8 s;
#pragma unroll
        for (int k = 0; k < 8; k++) s.v[k] = (uint8_t)o[k];
        *reinterpret_cast<StU8x8 *>(out + x0) = s;
      }
    } else {
      for (int k = 0; x0 + k < w; k++) {
        if (a.wide) {
          out[2 * (x0 + k)] = (uint8_t)(o[k] & 0xff);
          out[2 * (x0 + k) + 1] = (uint8_t)(o[k] >> 8);
        } else {
          out[x0 + k] = (uint8_t)o[k];
        }
      }
    }
  }
}

// ---- CRC-16 (x^16 + x^12 + x^5 + 1), the reference's bit-serial register ----
// The register after N message bits is (preset * x^N + M(x)) mod P, and the 16
// trailing zero bits multiply by x^16: linear over GF(2).  So every row's
// M_row(x) mod P is computed by its own wave, multiplied by x^(number of
// message bits that follow the row, + 16) and XOR-ed into the result word; the
// preset term is added by the wave of the first row (in mode 0 every plane's
// pieces already carry the bits of the planes after it, so the three plane
// words simply XOR together).  All powers of x come from
// a table of x^(2^i) mod P passed with the launch (host-computed constants).
#define XVC_CRC_POLY 0x1021u

struct CrcPow2 { uint16_t v[48]; };  // x^(2^i) mod P

__host__ __device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--) {
    r <<= 1;
    if (r & 0x10000u) r ^= 0x10000u | XVC_CRC_POLY;
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}

// x^n mod P, computed by a whole wave: lane i contributes x^(2^i) if bit i of n
// is set, then a 6-level product tree.  Every lane returns the result.
__device__ __forceinline__ uint32_t crc_xpow_wave(unsigned long long n, const CrcPow2 &t,
                                                  int lane) {
  uint32_t r = (lane < 48 && ((n >> lane) & 1ull)) ? t.v[lane] : 1u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) r = crc_mulmod(r, __shfl_xor(r, d, XVC_WAVE));
  return r;
}

struct CrcArgs {
  CrcPow2 pow2;
  int wide;   // 16 message bits per sample (bit depth > 8), else 8
  int mode;   // 0: one value over Y,U,V; 1: one per plane
};

// grid: (ceil(rows / 4), 3); block 256: one wave per row.  Lane t of a 64-lane
// window owns 16 samples; windows are aligned to the END of the row so that
// the incomplete one comes first, where missing samples are leading zeros and
// change nothing.  acc[c or 0] ^= contribution of the row.
// Byte-indexed tables, built once on the host for each sample width (they depend
// on nothing else): tab = h * x^16 mod P (the byte step of the register);
// mul_lo / mul_hi[l] = multiplication by a joining constant split by operand
// byte (GF(2)-linear: c * v = c * hi(v) * x^8 + c * lo(v)) - 2 LDS reads per
// product instead of a 16-step shift-and-xor loop.  l = 0..6: x^(lane bits << l),
// l = 7: x^(bits of 4 samples).
struct CrcTables {
  uint16_t tab[256];
  uint16_t mul_lo[8][256], mul_hi[8][256];
};

inline void crc_build_tables(CrcTables &t, const CrcPow2 &pow2, int wide) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t r = i << 8;
    for (int b = 0; b < 8; b++)
      r = (r & 0x8000u) ? ((r << 1) ^ XVC_CRC_POLY) & 0xffffu : (r << 1) & 0xffffu;
    t.tab[i] = (uint16_t)r;
    const int lb0 = wide ? 8 : 7;
    for (int l = 0; l < 8; l++) {
      const uint32_t c = pow2.v[l < 7 ? lb0 + l : lb0 - 2];
      t.mul_lo[l][i] = (uint16_t)crc_mulmod(c, i);
      t.mul_hi[l][i] = (uint16_t)crc_mulmod(c, i << 8);
    }
  }
}

__global__ void __launch_bounds__(256)
crc_rows_kernel(PicView pic, CrcArgs a, const CrcTables *tables, uint32_t *acc_words) {
  __shared__ CrcTables t;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(tables);
    uint4 *dst = reinterpret_cast<uint4 *>(&t);
    for (int i = threadIdx.x; i < (int)(sizeof(CrcTables) / 16); i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const uint16_t *tab = t.tab;
  const uint16_t (*mul_lo)[256] = t.mul_lo, (*mul_hi)[256] = t.mul_hi;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.y, y = blockIdx.x * 4 + (threadIdx.x >> 6);
  const PlaneView p = pic.c[c];
  if ((int)(blockIdx.x * 4) >= p.h) return;   // whole workgroup beyond the plane
  const bool live = y < p.h;
  const uint16_t *row = p.p + (ptrdiff_t)(live ? y : 0) * p.stride;
  const int w = p.w;
  const int n_win = live ? (w + 1023) / 1024 : 0;
  uint32_t acc = 0;
  for (int k = 0; k < n_win; k++) {
    const int x0 = w - (n_win - k) * 1024 + lane * 16;
    uint32_t r = 0;
    if (x0 + 16 > 0) {
      uint16_t v[16];
      if (x0 >= 0) {
        const StU16x8 s0 = *reinterpret_cast<const StU16x8 *>(row + x0);
        const StU16x8 s1 = *reinterpret_cast<const StU16x8 *>(row + x0 + 8);
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = s0.v[i]; v[8 + i] = s1.v[i]; }
      } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = x0 + i >= 0 ? row[x0 + i] : 0;
      }
      // four independent chains of four samples (the table walk is a chain of
      // dependent LDS reads), joined with x^(bits of 4 samples)
      uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const uint32_t sv = v[4 * c + i];
          q[c] = tab[q[c] >> 8] ^ ((q[c] & 0xffu) << 8) ^ (sv & 0xffu);
          if (a.wide) q[c] = tab[q[c] >> 8] ^ ((q[c] & 0xffu) << 8) ^ (sv >> 8);
        }
      }
      r = q[0];
#pragma unroll
      for (int c = 1; c < 4; c++) r = (uint32_t)(mul_hi[7][r >> 8] ^ mul_lo[7][r & 0xff]) ^ q[c];
    }
    // butterfly: after level l lane t (t % 2^(l+1) == 0) holds the piece of
    // 2^(l+1) lanes; the right half is x^(lane bits << l) = pow2[lb + l] shorter
#pragma unroll
    for (int l = 0; l < 6; l++) {
      const uint32_t right = __shfl_down(r, 1 << l, XVC_WAVE);
      r = (uint32_t)(mul_hi[l][r >> 8] ^ mul_lo[l][r & 0xff]) ^ right;
    }
    acc = (uint32_t)(mul_hi[6][acc >> 8] ^ mul_lo[6][acc & 0xff]) ^ r;  // lane 0's counts
  }
  acc = __shfl(acc, 0, XVC_WAVE);
  // message bits after this row (+ the 16 zero bits), and the preset term
  const unsigned long long bps = a.wide ? 16 : 8;
  unsigned long long after = (unsigned long long)w * (p.h - 1 - y) * bps + 16;
  unsigned long long before = (unsigned long long)w * y * bps;
  if (a.mode == 0) {
    for (int q = c + 1; q < 3; q++) after += (unsigned long long)pic.c[q].w * pic.c[q].h * bps;
    for (int q = 0; q < c; q++) before += (unsigned long long)pic.c[q].w * pic.c[q].h * bps;
  }
  uint32_t out = crc_mulmod(acc, crc_xpow_wave(after, a.pow2, lane));
  if (before == 0)  // first row of the message: preset * x^(N + 16)
    out ^= crc_mulmod(0xffffu, crc_xpow_wave(after + (unsigned long long)w * bps, a.pow2, lane));
  // one word per workgroup and plane (no same-address atomics: a few thousand
  // of those cost more than the whole computation); crc_finish_kernel XORs them
  __shared__ uint32_t wg_part[4];
  if (lane == 0) wg_part[threadIdx.x >> 6] = out;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t v = 0;
    for (int q = 0; q < 4; q++)
      if ((int)(blockIdx.x * 4 + q) < p.h) v ^= wg_part[q];
    acc_words[c * gridDim.x + blockIdx.x] = v;
  }
}

// grid 1; block 256: XOR of the per-workgroup words of each plane -> hash bytes
// (high byte first).  n_wg = gridDim.x of crc_rows_kernel.
__global__ void __launch_bounds__(256)
crc_finish_kernel(PicView pic, int mode, int n_wg, const uint32_t *acc_words, uint8_t *hash) {
  __shared__ uint32_t red[3][4];
  for (int c = 0; c < 3; c++) {
    const int live = (pic.c[c].h + 3) / 4;   // workgroups that wrote a word for plane c
    uint32_t v = 0;
    for (int i = threadIdx.x; i < live; i += 256) v ^= acc_words[c * n_wg + i];
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) v ^= __shfl_xor(v, s2, XVC_WAVE);
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r[3];
    for (int c = 0; c < 3; c++) r[c] = red[c][0] ^ red[c][1] ^ red[c][2] ^ red[c][3];
    if (mode == 0) r[0] ^= r[1] ^ r[2];
    for (int k = 0; k < (mode ? 3 : 1); k++) {
      hash[2 * k] = (uint8_t)(r[k] >> 8);
      hash[2 * k + 1] = (uint8_t)(r[k] & 0xff);
    }
  }
}

// ---- AQP variance statistic -------------------------------------------------
// grid: ceil(blocks / 4); block 256: one wave per 16x16 luma block, lane = 4
// consecutive samples.  out[by * bw + bx] = 256 * (sum(x^2) - sum(x)^2 / 256) / 256.
__global__ void __launch_bounds__(256)
variance_map_kernel(PlaneView p, int bw, int n_blocks, unsigned long long *out) {
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= n_blocks) return;
  const int lane = threadIdx.x & 63;
  const int bx = blk % bw, by = blk / bw;
  const uint16_t *s = p.p + (ptrdiff_t)(by * 16 + (lane >> 2)) * p.stride + bx * 16 + (lane & 3) * 4;
  const uint2 v = *reinterpret_cast<const uint2 *>(s);  // 8-byte aligned: x multiple of 4
  const uint32_t a0 = v.x & 0xffff, a1 = v.x >> 16, a2 = v.y & 0xffff, a3 = v.y >> 16;
  uint32_t sum = a0 + a1 + a2 + a3;
  uint32_t sq = a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;  // 256 * 4095^2 < 2^32
  sum = group_sum<64>(sum);
  sq = group_sum<64>(sq);
  if (lane == 0) {
    const unsigned long long su = sum, q = sq;
    out[blk] = (256ull * (q - (su * su) / 256ull)) / 256ull;
  }
}

// grid: ceil(n_ctus / 256); block 256: thread per CTU.  1 + the element
// [blocks / 2] of the sorted in-picture 16x16 variances of the CTU.
__global__ void __launch_bounds__(256)
ctu_variance_kernel(const unsigned long long *var_map, int w, int h, int ctu_size,
                    unsigned long long *out) {
  const int bw = (w + 15) / 16;
  const int cw = (w + ctu_size - 1) / ctu_size, chh = (h + ctu_size - 1) / ctu_size;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cw * chh) return;
  const int x = (i % cw) * ctu_size, y = (i / cw) * ctu_size, n = ctu_size / 16;
  unsigned long long v[64];
  int blocks = 0;
  for (int r = 0; r < n; r++) {
    if (y + r * 16 >= h) continue;
    for (int q = 0; q < n; q++) {
      if (x + q * 16 >= w) continue;
      const unsigned long long e = var_map[(y / 16 + r) * bw + x / 16 + q];
      int k = blocks++;
      while (k > 0 && v[k - 1] > e) { v[k] = v[k - 1]; k--; }
      v[k] = e;
    }
  }
  out[i] = 1ull + v[blocks / 2];
}

// ---- LIC histogram distance --------------------------------------------------
// grid: up to 512 workgroups over row slabs; block 256.  hist[value] += count in
// a - count in b (signed); a workgroup histograms in LDS first.
__global__ void __launch_bounds__(256)
histogram_diff_kernel(PlaneView a, PlaneView b, int buckets, int rows_per_wg, int *hist) {
  extern __shared__ int lh[];
  for (int i = threadIdx.x; i < buckets; i += 256) lh[i] = 0;
  __syncthreads();
  const int y0 = blockIdx.x * rows_per_wg;
  const int y1 = y0 + rows_per_wg < a.h ? y0 + rows_per_wg : a.h;
  const int cpr = a.w >> 3;  // 16-byte chunks per row
  for (int i = threadIdx.x; i < (y1 - y0) * cpr; i += 256) {
    const int y = y0 + i / cpr, x = (i % cpr) << 3;
    const uint4 va = *reinterpret_cast<const uint4 *>(a.p + (ptrdiff_t)y * a.stride + x);
    const uint4 vb = *reinterpret_cast<const uint4 *>(b.p + (ptrdiff_t)y * b.stride + x);
    const uint32_t ua[4] = {va.x, va.y, va.z, va.w}, ub[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      atomicAdd(&lh[(ua[k] & 0xffff) & (buckets - 1)], 1);
      atomicAdd(&lh[(ua[k] >> 16) & (buckets - 1)], 1);
      atomicAdd(&lh[(ub[k] & 0xffff) & (buckets - 1)], -1);
      atomicAdd(&lh[(ub[k] >> 16) & (buckets - 1)], -1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < buckets; i += 256)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// grid 1; block 256: out = sum |hist|; hist is cleared for the next call.
__global__ void __launch_bounds__(256)
histogram_abs_sum_kernel(int *hist, int buckets, long long *out) {
  __shared__ long long part[4];
  long long s = 0;
  for (int i = threadIdx.x; i < buckets; i += 256) {
    const int v = hist[i];
    hist[i] = 0;
    s += v < 0 ? -(long long)v : v;
  }
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = part[0] + part[1] + part[2] + part[3];
}

#endif  // XVCGPU_K_STATS_H_
