// k_tx.h -- X1, Q (QuantFast), Q1, X2, R1: one workgroup = one call of
// TransformEncoder::TransformAndReconstruct (transform_encoder.cc:203-285)
// for one (CU, component).
//
// Every 1-D pass is the plain matrix product with int32 accumulation that the
// reference's partial butterflies regroup; int32 wrap-
// around is a ring homomorphism so the regrouping is bit-identical even on
// overflow.  Matrices (8-bit fraction, values up to +-362: no int8 MFMA fit)
// are staged in LDS; blocks use a row stride of 66 int16 (33 dwords) so the
// column walks of the forward passes are bank-conflict free.
//   forward : horizontal (width-point, no zero-out) then vertical (zero-out)
//   inverse : vertical first (zero-out) then horizontal, clip16 after each
// 64-point transforms keep only the 32 low-frequency outputs / inputs.
#ifndef XVCGPU_K_TX_H_
#define XVCGPU_K_TX_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "xvcgpu_internal.h"

#define TX_THREADS 256
#define TX_S 66  // LDS row stride in int16

enum { TX_MODE_FULL = 0, TX_MODE_FWD = 1, TX_MODE_INV = 2 };

struct TxShared {
  int16_t a[64 * TX_S];
  int16_t b[64 * TX_S];
  int16_t mh[64 * 64];
  int16_t mv[64 * 64];
  int nnz;
};

__device__ __forceinline__ int tx_table_off(const TxTableLayout &lay, int type,
                                            int size) {
  if (type == XVC_TX_DEFAULT) type = XVC_TX_DCT2;
  return lay.off[type - 1][d_log2_size(size)];
}

// out[k*S + y] = (sum_j M[k*N + j] * in[y*S + j] + add) >> shift, unclipped;
// k < min(N,32), y < tx_lines; zero elsewhere inside N x lines.
__device__ __forceinline__ void tx_fwd_1d(const int16_t *M, int N, int shift,
                                          int lines, bool zero_out,
                                          const int16_t *in, int16_t *out) {
  const int add = 1 << (shift - 1);
  const int tx_lines = zero_out ? min(lines, 32) : lines;
  const int out_rows = min(N, 32);
  const int ll = 31 - __clz(lines);  // lines is a power of two
  for (int i = threadIdx.x; i < N * lines; i += TX_THREADS) {
    const int k = i >> ll, y = i & (lines - 1);
    int16_t v = 0;
    if (k < out_rows && y < tx_lines) {
      int sum = 0;
      const int16_t *m = M + k * N;
      const int16_t *p = in + y * TX_S;
      for (int j = 0; j < N; j++) sum += (int)m[j] * (int)p[j];
      v = (int16_t)((sum + add) >> shift);
    }
    out[k * TX_S + y] = v;
  }
}

// out[y*S + k] = clip16((sum_{j<min(N,32)} M[j*N + k] * in[j*S + y] + add) >>
// shift) for y < tx_lines; zero rows beyond.
__device__ __forceinline__ void tx_inv_1d(const int16_t *M, int N, int shift,
                                          int lines, bool zero_out,
                                          const int16_t *in, int16_t *out) {
  const int add = 1 << (shift - 1);
  const int tx_lines = zero_out ? min(lines, 32) : lines;
  const int in_rows = min(N, 32);
  const int ln = 31 - __clz(N);
  for (int i = threadIdx.x; i < N * lines; i += TX_THREADS) {
    const int y = i >> ln, k = i & (N - 1);
    int16_t v = 0;
    if (y < tx_lines) {
      int sum = 0;
      for (int j = 0; j < in_rows; j++)
        sum += (int)M[j * N + k] * (int)in[j * TX_S + y];
      v = (int16_t)d_clip3((sum + add) >> shift, -32768, 32767);
    }
    out[y * TX_S + k] = v;
  }
}

// FwdPartialDst4 / InvPartialDst4 (transform.cc:997-1017, :217-242); `shift`
// already reduced by the high-precision delta.  4 threads, one line each.
__device__ __forceinline__ void tx_fwd_dst4(int shift, const int16_t *in,
                                            int16_t *out) {
  const int i = threadIdx.x;
  if (i < 4) {
    const int add = 1 << (shift - 1);
    const int16_t *p = in + i * TX_S;
    const int c0 = p[0] + p[3], c1 = p[1] + p[3], c2 = p[0] - p[1], c3 = 74 * p[2];
    out[0 * TX_S + i] = (int16_t)((29 * c0 + 55 * c1 + c3 + add) >> shift);
    out[1 * TX_S + i] = (int16_t)((74 * (p[0] + p[1] - p[3]) + add) >> shift);
    out[2 * TX_S + i] = (int16_t)((29 * c2 + 55 * c0 - c3 + add) >> shift);
    out[3 * TX_S + i] = (int16_t)((55 * c2 - 29 * c1 + c3 + add) >> shift);
  }
}
__device__ __forceinline__ void tx_inv_dst4(int shift, const int16_t *in,
                                            int16_t *out) {
  const int i = threadIdx.x;
  if (i < 4) {
    const int add = 1 << (shift - 1);
    const int i0 = in[0 * TX_S + i], i1 = in[1 * TX_S + i], i2 = in[2 * TX_S + i],
              i3 = in[3 * TX_S + i];
    const int c0 = i0 + i2, c1 = i2 + i3, c2 = i0 - i3, c3 = 74 * i1;
    int16_t *o = out + i * TX_S;
    o[0] = (int16_t)d_clip3((29 * c0 + 55 * c1 + c3 + add) >> shift, -32768, 32767);
    o[1] = (int16_t)d_clip3((55 * c2 - 29 * c1 + c3 + add) >> shift, -32768, 32767);
    o[2] = (int16_t)d_clip3((74 * (i0 - i2 + i3) + add) >> shift, -32768, 32767);
    o[3] = (int16_t)d_clip3((55 * c0 + 29 * c2 - c3 + add) >> shift, -32768, 32767);
  }
}

// Jobs taken by the one-wave-per-job kernel (k_tx2.h); the rest stay here.
__device__ __forceinline__ bool tx_small_job(const xvcgpu_tx_block &b) {
  const bool okw = b.w == 4 || b.w == 8 || b.w == 16;
  const bool okh = b.h == 4 || b.h == 8 || b.h == 16;
  return okw && okh && !(b.dst4x4 && b.w == 4 && b.h == 4) && b.tx_hor != XVC_TX_SKIP;
}

// One workgroup (256 threads) = one job: the general path (blocks above
// 16x16, 2-wide blocks, the 4x4 DST).
template <int MODE>
__device__ __forceinline__ void residual_job(TxShared &s, int bi, const PicView &orig,
                                             const PicView &pred, const PicView &rec,
                                             const xvcgpu_tx_block *blocks,
                                             int16_t *levels, const uint32_t *level_off,
                                             int32_t *nnz_out, const int16_t *tx_tables,
                                             const TxTableLayout &lay) {
  __syncthreads();  // previous job of this workgroup is done with s
  const xvcgpu_tx_block b = blocks[bi];
  const int w = b.w, h = b.h, bd = pred.bd;
  const int lw = 31 - __clz(w);
  const int lgw = d_log2_size(w), lgh = d_log2_size(h);
  const PlaneView pp = pred.c[b.comp];
  const bool skip = b.tx_hor == XVC_TX_SKIP;  // TransformSkip, blocks <= 4x4
  const bool dst4 = b.dst4x4 && w == 4 && h == 4 && !skip;
  int16_t *lv = (levels && level_off) ? levels + level_off[bi] : nullptr;

  // matrices
  if (!dst4 && !skip) {
    const int16_t *gh = tx_tables + tx_table_off(lay, b.tx_hor, w);
    const int16_t *gv = tx_tables + tx_table_off(lay, b.tx_ver, h);
    for (int i = threadIdx.x; i < w * w; i += TX_THREADS) s.mh[i] = gh[i];
    for (int i = threadIdx.x; i < h * h; i += TX_THREADS) s.mv[i] = gv[i];
  }
  if (threadIdx.x == 0) s.nnz = 0;

  // quantiser parameters (quantize.cc:48-72, :94-131; rdo_quant.cc:160-169)
  int qpb = b.qp + 6 * (bd - 8);
  qpb = qpb > 0 ? qpb : 0;
  const bool bias = ((lgw + lgh) & 1) != 0;
  const int tshift = 15 - bd - ((lgw + lgh) >> 1);

  int nnz;
  if (MODE != TX_MODE_INV) {
    // residual (ResidualBuffer::Subtract, sample_buffer.h:130-145)
    const PlaneView po = orig.c[b.comp];
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int o = po.p[(ptrdiff_t)(b.y + y) * po.stride + b.x + x];
      const int p = pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
      s.a[y * TX_S + x] = (int16_t)(o - p);
    }
    __syncthreads();
    // forward transform (transform.cc:869-961, high precision)
    const int shift1 = lgw + bd - 9 + 2, shift2 = lgh + 6 + 2;
    if (skip) {  // ForwardTransform::TransformSkip, transform.cc:963-995
      const int sh = tshift + (bias ? -8 : 0), sc = bias ? 181 : 1;
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
        const int k = (i >> lw) * TX_S + (i & (w - 1));
        const int v = (int)s.a[k] * sc;
        s.a[k] = sh > 0 ? (int16_t)(v * (1 << sh)) : (int16_t)((v + (1 << (-sh - 1))) >> -sh);
      }
    } else if (dst4) {
      tx_fwd_dst4(shift1 - 2, s.a, s.b);
      __syncthreads();
      tx_fwd_dst4(shift2 - 2, s.b, s.a);
    } else {
      tx_fwd_1d(s.mh, w, shift1, h, false, s.a, s.b);
      __syncthreads();
      tx_fwd_1d(s.mv, h, shift2, w, true, s.b, s.a);
    }
    __syncthreads();
    // s.a[ky*S + kx] = coefficients
    if (MODE == TX_MODE_FWD) {
      if (lv)
        for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
          lv[i] = s.a[(i >> lw) * TX_S + (i & (w - 1))];
      return;
    }
    // QuantFast (rdo_quant.cc:156-195)
    const int qshift = 14 + qpb / 6 + tshift + (bias ? 7 : 0);
    const int qscale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
    const long long qoff = (long long)((b.intra_pic ? 171ull : 85ull) << (qshift - 9));
    int local = 0;
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int v = s.a[y * TX_S + x];
      const int sign = v < 0 ? -1 : 1;
      const long long abs_coeff = d_abs(v);
      const int level = (int)(((abs_coeff * qscale) + qoff) >> qshift);
      local += level != 0;
      const int16_t q = (int16_t)d_clip3(level * sign, -32768, 32767);
      s.a[y * TX_S + x] = q;
      if (lv) lv[i] = q;
    }
    local = group_sum<64>(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s.nnz, local);
    __syncthreads();
    nnz = s.nnz;
    if (nnz_out && threadIdx.x == 0) nnz_out[bi] = nnz;
  } else {
    nnz = nnz_out[bi];
    if (nnz)
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
        s.a[(i >> lw) * TX_S + (i & (w - 1))] = lv[i];
    __syncthreads();
  }

  const PlaneView pr = rec.c[b.comp];
  if (nnz == 0) {  // cbf == 0: rec = pred (CopyFrom, transform_encoder.cc:281)
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      pr.p[(ptrdiff_t)(b.y + y) * pr.stride + b.x + x] =
          pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
    }
    return;
  }
  const bool dc_only = nnz == 1 && s.a[0] != 0;  // transform_encoder.cc:241

  // Quantize::Inverse (quantize.cc:94-125), in place in s.a
  {
    const int shift = 6 - tshift + (bias ? 8 : 0);
    const int scale = (kInvQuantScales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
    __syncthreads();  // dc_only read s.a[0]
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int prod = (int)s.a[y * TX_S + x] * scale;
      int cf;
      if (shift > 0)
        cf = (prod + (1 << (shift - 1))) >> shift;
      else
        cf = (int)((unsigned)prod << -shift);
      s.a[y * TX_S + x] = (int16_t)d_clip3(cf, -32768, 32767);
    }
    __syncthreads();
  }
  // InverseTransform::Transform (transform.cc:83-182) -> residual in s.a
  {
    const int shift1 = 7 + 2, shift2 = 20 - bd + 2;
    const bool dct2_both =
        (b.tx_ver == XVC_TX_DEFAULT || b.tx_ver == XVC_TX_DCT2) &&
        (b.tx_hor == XVC_TX_DEFAULT || b.tx_hor == XVC_TX_DCT2);
    if (skip) {  // InverseTransform::TransformSkip, transform.cc:184-215
      const int sh = tshift + (bias ? 7 : 0), sc = bias ? 181 : 1;
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
        const int k = (i >> lw) * TX_S + (i & (w - 1));
        const int v = (int)s.a[k] * sc;
        s.a[k] = sh > 0 ? (int16_t)((v + (1 << (sh - 1))) >> sh)
                        : (int16_t)((uint32_t)v << -sh);
      }
    } else if (dst4) {
      tx_inv_dst4(shift1 - 2, s.a, s.b);
      __syncthreads();
      tx_inv_dst4(shift2 - 2, s.b, s.a);
    } else if (dc_only && dct2_both) {  // InvDct2Dc, transform.cc:279-291
      const int sh = 14 - bd, add = 1 << (sh - 1);
      const int16_t cf = (int16_t)(((((int)s.a[0] + 1) >> 1) + add) >> sh);
      __syncthreads();
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
        s.a[(i >> lw) * TX_S + (i & (w - 1))] = cf;
    } else {
      tx_inv_1d(s.mv, h, shift1, w, true, s.a, s.b);
      __syncthreads();
      tx_inv_1d(s.mh, w, shift2, h, false, s.b, s.a);
    }
    __syncthreads();
  }
  // SampleBuffer::AddClip (sample_buffer.h:72-87)
  const int smax = (1 << bd) - 1;
  for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
    const int y = i >> lw, x = i & (w - 1);
    const int p = pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
    pr.p[(ptrdiff_t)(b.y + y) * pr.stride + b.x + x] =
        (uint16_t)d_clip3(p + (int)s.a[y * TX_S + x], 0, smax);
  }
}

// grid: ceil(n/256) workgroups of 256 threads.  Each workgroup scans 256
// descriptors, collects the jobs that need the general path and runs them one
// after the other (normally none: every job of a 16x16-CU picture is "small").
template <int MODE>
__global__ void __launch_bounds__(TX_THREADS)
residual_kernel(PicView orig, PicView pred, PicView rec,
                const xvcgpu_tx_block *blocks, int n, int16_t *levels,
                const uint32_t *level_off, int32_t *nnz_out,
                const int16_t *tx_tables, TxTableLayout lay) {
  __shared__ __attribute__((aligned(16))) TxShared s;
  __shared__ int jobs[TX_THREADS];
  __shared__ int n_jobs;
  if (threadIdx.x == 0) n_jobs = 0;
  __syncthreads();
  const int idx = blockIdx.x * TX_THREADS + threadIdx.x;
  if (idx < n && !tx_small_job(blocks[idx])) jobs[atomicAdd(&n_jobs, 1)] = idx;
  __syncthreads();
  const int nj = n_jobs;
  for (int k = 0; k < nj; k++)
    residual_job<MODE>(s, jobs[k], orig, pred, rec, blocks, levels, level_off, nnz_out,
                       tx_tables, lay);
}

#endif  // XVCGPU_K_TX_H_
