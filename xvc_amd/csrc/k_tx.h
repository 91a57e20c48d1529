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
  int16_t dl[64 * TX_S];  // QuantFast rounding remainders (sign-data hiding)
  int nnz;
  int last_sb;
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

// ---- sign-data hiding: RdoQuant::CoeffSignHideFast (rdo_quant.cc:448-573) --
// Scan position k of a 4x4 sub-block as y*4 + x (TransformHelper::
// kScanCoeff4x4, transform.cc:72-76): order 0 walks the anti-diagonals from
// bottom-left to top-right, 1 is raster, 2 is column-major.
constexpr unsigned long long tx_pack_scan4(int order) {
  unsigned long long t = 0;
  int k = 0;
  if (order == 0) {
    for (int s = 0; s < 7; s++)
      for (int y = (s < 4 ? s : 3); y >= 0 && s - y < 4; y--, k++)
        t |= (unsigned long long)((y << 2) | (s - y)) << (4 * k);
  } else {
    for (k = 0; k < 16; k++)
      t |= (unsigned long long)(order == 1 ? k : (((k & 3) << 2) | (k >> 2))) << (4 * k);
  }
  return t;
}
// 16 nibbles: scan position k -> y*4 + x
__device__ __forceinline__ unsigned long long d_scan4_table(int order) {
  constexpr unsigned long long t0 = tx_pack_scan4(0), t1 = tx_pack_scan4(1),
                               t2 = tx_pack_scan4(2);
  return order == 0 ? t0 : (order == 1 ? t1 : t2);
}
// Index of sub-block (sx, sy) in the scan over a gw x gh grid of sub-blocks
// (TransformHelper::DeriveSubblockScan, transform.cc:1639-1683).
__device__ __forceinline__ int d_sb_scan_index(int order, int gw, int gh, int sx, int sy) {
  if (order == 1) return sy * gw + sx;
  if (order == 2) return sx * gh + sy;
  const int s = sx + sy;
  int idx = 0;
  for (int d = 0; d < s; d++) {
    int c = d < gw - 1 ? d : gw - 1;
    c = c < gh - 1 ? c : gh - 1;
    c = c < gw + gh - 2 - d ? c : gw + gh - 2 - d;
    idx += c + 1;
  }
  return idx + ((s < gh - 1 ? s : gh - 1) - sy);
}

// One 4x4 sub-block, by one thread.  IDX(x, y) maps a coefficient position
// to the index used by the three arrays (levels in/out, remainders, the
// unquantised coefficients).  Returns the change of the non-zero count.
template <typename IDX>
__device__ __forceinline__ int d_sign_hide_subblock(int order, int px, int py,
                                                    bool is_last_sb, int16_t *lev,
                                                    const int16_t *dl, const int16_t *cf,
                                                    IDX idx) {
  const unsigned long long tab = d_scan4_table(order);
  auto at = [&](int k) {
    const int p = (int)((tab >> (4 * k)) & 15ull);
    return idx(px + (p & 3), py + (p >> 2));
  };
  int last = -1, first = 16, sum = 0;
  for (int k = 0; k < 16; k++) {
    const int c = lev[at(k)];
    if (c) {
      first = k < first ? k : first;
      last = k;
      sum += c;
    }
  }
  if (last - first <= 3) return 0;
  const int sign = lev[at(first)] > 0 ? 0 : 1;
  if (sign == (sum & 1)) return 0;
  int curr_cost = 32767, curr_change = 0, min_cost = 32767, min_change = 0, min_index = 0;
  for (int k = is_last_sb ? last : 15; k >= 0; k--) {
    const int p = at(k);
    const int l = lev[p], d = dl[p];
    if (l != 0) {
      if (d > 0) { curr_cost = -d; curr_change = 1; }
      else if (k == first && d_abs(l) == 1) curr_cost = 32767;
      else { curr_cost = d; curr_change = -1; }
    } else if (k < first && (cf[p] >= 0 ? 0 : 1) != sign) {
      curr_cost = 32767;
    } else {
      curr_cost = -d;
      curr_change = 1;
    }
    if (curr_cost < min_cost) { min_cost = curr_cost; min_change = curr_change; min_index = k; }
  }
  const int p = at(min_index);
  const int before = lev[p];
  if (before == -32768 || before == 32767) min_change = -1;
  const int after = (int16_t)(before + (cf[p] >= 0 ? min_change : -min_change));
  lev[p] = (int16_t)after;
  return (after != 0) - (before != 0);
}

#include "k_rdoq.h"  // needs the scan helpers above

// Jobs taken by the one-wave-per-job kernel (k_tx2.h); the rest stay here.
__device__ __forceinline__ bool tx_small_job(const xvcgpu_tx_block &b) {
  const bool okw = b.w == 4 || b.w == 8 || b.w == 16;
  const bool okh = b.h == 4 || b.h == 8 || b.h == 16;
  return okw && okh && !(b.dst4x4 && b.w == 4 && b.h == 4) && b.tx_hor != XVC_TX_SKIP;
}

// One workgroup (256 threads) = one job: the general path (blocks above
// 16x16, 2-wide blocks, the 4x4 DST).
template <int MODE, int RQN = 4>
__device__ __forceinline__ int residual_job(TxShared &s, int bi, const PicView &orig,
                                             const PicView &pred, const PicView &rec,
                                             const xvcgpu_tx_block *blocks,
                                             int16_t *levels, const uint32_t *level_off,
                                             int32_t *nnz_out, const int16_t *tx_tables,
                                             const TxTableLayout &lay,
                                             RdoqShared<RQN> *rq = nullptr,
                                             const xvcgpu_rdoq_contexts *rq_ctx = nullptr,
                                             const xvcgpu_rdoq_params *rq_prm = nullptr,
                                             unsigned long long *dist_out = nullptr,
                                             const uint16_t *pred_blk = nullptr,
                                             int pred_blk_stride = 0,
                                             bool defer_add = false) {
  // pred_blk: the block's prediction handed over directly (row stride
  // pred_blk_stride, e.g. in LDS) instead of read from the `pred` picture.
  // defer_add (TX_MODE_INV): stop in front of AddClip and return the block's
  // level count - 0: no residual; else the residual is s.a[y * TX_S + x] - so
  // that a caller whose prediction is not ready yet adds it itself.  Returns -1
  // otherwise.
  __shared__ unsigned long long s_dist;  // see tx2_job: the residual-domain SSD
  if (threadIdx.x == 0) s_dist = 0;
  __syncthreads();  // previous job of this workgroup is done with s
  const xvcgpu_tx_block b = blocks[bi];
  const int w = b.w, h = b.h, bd = pred.bd;
  const int lw = 31 - __clz(w);
  const int lgw = d_log2_size(w), lgh = d_log2_size(h);
  const PlaneView pp = pred.c[b.comp];
  const bool skip = b.tx_hor == XVC_TX_SKIP;  // TransformSkip, blocks <= 4x4
  const bool intra_pic = (b.intra_pic & XVC_TXF_INTRA_PIC) != 0;
  const bool sign_hide = !(b.intra_pic & XVC_TXF_NO_SIGN_HIDING);
  const int scan_order = (b.intra_pic >> XVC_TXF_SCAN_SHIFT) & 3;
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
      const int p = pred_blk ? pred_blk[y * pred_blk_stride + x]
                             : pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
      s.a[y * TX_S + x] = (int16_t)(o - p);
    }
    __syncthreads();
    // forward transform (transform.cc:869-961; a stage of XVC_TX_DCT2_LOW has the
    // 6-bit matrix and no high-precision shift, :876-884)
    const int shift1 = lgw + bd - 9 + (b.tx_hor == XVC_TX_DCT2_LOW ? 0 : 2);
    const int shift2 = lgh + 6 + (b.tx_ver == XVC_TX_DCT2_LOW ? 0 : 2);
    if (skip) {  // ForwardTransform::TransformSkip, transform.cc:963-995
      const int sh = tshift + (bias ? -8 : 0), sc = bias ? 181 : 1;
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
        const int k = (i >> lw) * TX_S + (i & (w - 1));
        const int v = (int)s.a[k] * sc;
        s.a[k] = sh > 0 ? (int16_t)(v * (1 << sh)) : (int16_t)((v + (1 << (-sh - 1))) >> -sh);
      }
    } else if (dst4) {
      tx_fwd_dst4(lgw + bd - 9, s.a, s.b);
      __syncthreads();
      tx_fwd_dst4(lgh + 6, s.b, s.a);
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
      return -1;
    }
    // QuantFast (rdo_quant.cc:156-195)
    const int qshift = 14 + qpb / 6 + tshift + (bias ? 7 : 0);
    const int qscale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
    const long long qoff = (long long)((intra_pic ? 171ull : 85ull) << (qshift - 9));
    // levels -> s.b, rounding remainders -> s.dl, coefficients stay in s.a
    bool use_rdoq = RQN > 4 && (b.intra_pic & XVC_TXF_RDOQ) != 0;
    if (use_rdoq && (w == 2 || h == 2) && (rq_prm[bi].flags & XVC_RDOQ_NO_2X2))
      use_rdoq = false;  // rdo_quant.cc:208-216: 2-wide blocks fall back to QuantFast
    if (RQN > 4 && use_rdoq) {
      // RdoQuant::QuantRdo (rdo_quant.cc:203-446) by the first wave
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
        s.b[(i >> lw) * TX_S + (i & (w - 1))] = 0;  // levels beyond the 32x32 corner
      __syncthreads();
      if (threadIdx.x < 64) {
        const xvcgpu_rdoq_params prm = rq_prm[bi];
        const int16_t *cfp = s.a;
        int16_t *lvp = s.b;
        auto cf_at = [cfp](int x, int y) { return (int)cfp[y * TX_S + x]; };
        auto lv_at = [lvp](int x, int y) { return lvp + y * TX_S + x; };
        int n_rq;
        if (RQN >= 1024 && rq4_takes(64, w, h, scan_order)) {
          // four lanes per sub-block (k_rdoq4.h), four units per lane beyond sixteen sub-blocks
          rq_stage_costs(&rq_ctx[prm.ctx_index], rq->ctx_bits, (int)threadIdx.x, 64);
          wave_sync();
          if (rq4_needs_nr4(w, h))
            n_rq = wave_rdoq4<64, 4>(*rq, (int)threadIdx.x, bd, w, h, b.qp, b.comp == 0, sign_hide,
                                     prm, cf_at, lv_at);
          else
            n_rq = wave_rdoq4<64, 1>(*rq, (int)threadIdx.x, bd, w, h, b.qp, b.comp == 0, sign_hide,
                                     prm, cf_at, lv_at);
        } else {
          n_rq = wave_rdoq<64>(*rq, (int)threadIdx.x, bd, w, h, b.qp, b.comp == 0, scan_order,
                               sign_hide, rq_ctx[prm.ctx_index], prm, cf_at, lv_at);
        }
        if (threadIdx.x == 0) s.nnz = n_rq;
      }
      __syncthreads();
    }
    int local = 0;
    if (!use_rdoq)
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int v = s.a[y * TX_S + x];
      const int sign = v < 0 ? -1 : 1;
      const long long abs_coeff = d_abs(v);
      const int level = (int)(((abs_coeff * qscale) + qoff) >> qshift);
      local += level != 0;
      s.b[y * TX_S + x] = (int16_t)d_clip3(level * sign, -32768, 32767);
      s.dl[y * TX_S + x] =
          (int16_t)(((abs_coeff * qscale) - ((long long)level << qshift)) >> (qshift - 8));
    }
    local = group_sum<64>(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s.nnz, local);
    if (threadIdx.x == 0) s.last_sb = -1;
    __syncthreads();
    // CoeffSignHideFast (rdo_quant.cc:196-199, :448-573): thread = sub-block
    if (!use_rdoq && sign_hide && s.nnz > 1 && w >= 4 && h >= 4) {
      const int gw = w >> 2, gh = h >> 2;
      auto idx = [](int x, int y) { return y * TX_S + x; };
      // the "last" sub-block = highest scan index holding a non-zero level
      for (int t = threadIdx.x; t < gw * gh; t += TX_THREADS) {
        const int sx = t % gw, sy = t / gw;
        bool any = false;
        for (int k = 0; k < 16; k++) any |= s.b[idx(4 * sx + (k & 3), 4 * sy + (k >> 2))] != 0;
        if (any) atomicMax(&s.last_sb, d_sb_scan_index(scan_order, gw, gh, sx, sy));
      }
      __syncthreads();
      int dn = 0;
      for (int t = threadIdx.x; t < gw * gh; t += TX_THREADS) {
        const int sx = t % gw, sy = t / gw;
        dn += d_sign_hide_subblock(
            scan_order, 4 * sx, 4 * sy,
            d_sb_scan_index(scan_order, gw, gh, sx, sy) == s.last_sb, s.b, s.dl, s.a, idx);
      }
      dn = group_sum<64>(dn);
      if ((threadIdx.x & 63) == 0 && dn) atomicAdd(&s.nnz, dn);
      __syncthreads();
    }
    nnz = s.nnz;
    if (nnz_out && threadIdx.x == 0) nnz_out[bi] = nnz;
    if (lv)
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
        lv[i] = s.b[(i >> lw) * TX_S + (i & (w - 1))];
  } else {
    nnz = nnz_out[bi];
    if (nnz)
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS)
        s.b[(i >> lw) * TX_S + (i & (w - 1))] = lv[i];
    __syncthreads();
  }

  const PlaneView pr = rec.c[b.comp];
  const PlaneView pod = orig.c[b.comp];
  unsigned long long dist_acc = 0;
  auto dist_finish = [&]() {
    unsigned long long v = dist_acc;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_dist, v);
    __syncthreads();
    if (threadIdx.x == 0) dist_out[bi] = s_dist >> (2 * (bd - 8));
  };
  if (defer_add && nnz == 0) return 0;
  if (nnz == 0) {  // cbf == 0: rec = pred (CopyFrom, transform_encoder.cc:281)
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int p = pred_blk ? pred_blk[y * pred_blk_stride + x]
                             : pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
      pr.p[(ptrdiff_t)(b.y + y) * pr.stride + b.x + x] = (uint16_t)p;
      if (dist_out) {
        const int d = (int)pod.p[(ptrdiff_t)(b.y + y) * pod.stride + b.x + x] - p;
        dist_acc += (unsigned long long)((long long)d * d);
      }
    }
    if (dist_out) dist_finish();
    return -1;
  }
  const bool dc_only = nnz == 1 && s.b[0] != 0;  // transform_encoder.cc:241

  // Quantize::Inverse (quantize.cc:94-125): levels in s.b -> s.a
  {
    const int shift = 6 - tshift + (bias ? 8 : 0);
    const int scale = (kInvQuantScales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
    for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
      const int y = i >> lw, x = i & (w - 1);
      const int prod = (int)s.b[y * TX_S + x] * scale;
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
    const int shift1 = 7 + (b.tx_ver == XVC_TX_DCT2_LOW ? 0 : 2);
    const int shift2 = 20 - bd + (b.tx_hor == XVC_TX_DCT2_LOW ? 0 : 2);
    const bool dct2_both =
        (b.tx_ver == XVC_TX_DEFAULT || b.tx_ver == XVC_TX_DCT2 || b.tx_ver == XVC_TX_DCT2_LOW) &&
        (b.tx_hor == XVC_TX_DEFAULT || b.tx_hor == XVC_TX_DCT2 || b.tx_hor == XVC_TX_DCT2_LOW);
    if (skip) {  // InverseTransform::TransformSkip, transform.cc:184-215
      const int sh = tshift + (bias ? 7 : 0), sc = bias ? 181 : 1;
      for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
        const int k = (i >> lw) * TX_S + (i & (w - 1));
        const int v = (int)s.a[k] * sc;
        s.a[k] = sh > 0 ? (int16_t)((v + (1 << (sh - 1))) >> sh)
                        : (int16_t)((uint32_t)v << -sh);
      }
    } else if (dst4) {
      tx_inv_dst4(7, s.a, s.b);
      __syncthreads();
      tx_inv_dst4(20 - bd, s.b, s.a);
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
  if (defer_add) return nnz;
  // SampleBuffer::AddClip (sample_buffer.h:72-87)
  const int smax = (1 << bd) - 1;
  for (int i = threadIdx.x; i < w * h; i += TX_THREADS) {
    const int y = i >> lw, x = i & (w - 1);
    const int p = pred_blk ? pred_blk[y * pred_blk_stride + x]
                           : pp.p[(ptrdiff_t)(b.y + y) * pp.stride + b.x + x];
    pr.p[(ptrdiff_t)(b.y + y) * pr.stride + b.x + x] =
        (uint16_t)d_clip3(p + (int)s.a[y * TX_S + x], 0, smax);
    if (dist_out) {
      const int d = (int)pod.p[(ptrdiff_t)(b.y + y) * pod.stride + b.x + x] - p -
                    (int)s.a[y * TX_S + x];
      dist_acc += (unsigned long long)((long long)d * d);
    }
  }
  if (dist_out) dist_finish();
  return -1;
}

// grid: ceil(n/256) workgroups of 256 threads.  Each workgroup scans 256
// descriptors, collects the jobs that need the general path and runs them one
// after the other (normally none: every job of a 16x16-CU picture is "small").
template <int MODE, bool RDOQ>
__device__ __forceinline__ void residual_kernel_body(PicView orig, PicView pred, PicView rec, const xvcgpu_tx_block *blocks, int n, int16_t *levels, const uint32_t *level_off, int32_t *nnz_out, const int16_t *tx_tables, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, unsigned long long *dist_out) {
  __shared__ __attribute__((aligned(16))) TxShared s;
  __shared__ RdoqShared<RDOQ ? 1024 : 4> rq;
  __shared__ int jobs[TX_THREADS];
  __shared__ int n_jobs;
  if (threadIdx.x == 0) n_jobs = 0;
  __syncthreads();
  const int idx = blockIdx.x * TX_THREADS + threadIdx.x;
  if (idx < n && !tx_small_job(blocks[idx])) jobs[atomicAdd(&n_jobs, 1)] = idx;
  __syncthreads();
  const int nj = n_jobs;
  for (int k = 0; k < nj; k++)
    residual_job<MODE, RDOQ ? 1024 : 4>(s, jobs[k], orig, pred, rec, blocks, levels, level_off,
                                        nnz_out, tx_tables, lay, &rq, rq_ctx, rq_prm, dist_out);
}

template <int MODE, bool RDOQ = false>
__global__ void __launch_bounds__(TX_THREADS)
residual_kernel(PicView orig, PicView pred, PicView rec, const xvcgpu_tx_block *blocks, int n, int16_t *levels, const uint32_t *level_off, int32_t *nnz_out, const int16_t *tx_tables, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx = nullptr, const xvcgpu_rdoq_params *rq_prm = nullptr, unsigned long long *dist_out = nullptr) {
  residual_kernel_body<MODE, RDOQ>(orig, pred, rec, blocks, n, levels, level_off, nnz_out, tx_tables, lay, rq_ctx, rq_prm, dist_out);
}

// The same path with one workgroup per descriptor (a workgroup whose block
// belongs to the one-wave kernel retires at once): for batches made mostly of
// large blocks - the dependency waves of a decoded picture hold a few dozen
// 32x32 / 64x64 blocks each, which the scanning form above would run one after
// the other in a single workgroup.  grid: n; block: TX_THREADS.
template <int MODE, bool RDOQ = false>
__global__ void __launch_bounds__(TX_THREADS)
residual_per_job_kernel(PicView orig, PicView pred, PicView rec,
                        const xvcgpu_tx_block *blocks, int n, int16_t *levels,
                        const uint32_t *level_off, int32_t *nnz_out,
                        const int16_t *tx_tables, TxTableLayout lay,
                        unsigned long long *dist_out = nullptr,
                        const xvcgpu_rdoq_contexts *rq_ctx = nullptr,
                        const xvcgpu_rdoq_params *rq_prm = nullptr) {
  __shared__ __attribute__((aligned(16))) TxShared s;
  __shared__ RdoqShared<RDOQ ? 1024 : 4> rq;
  const int idx = blockIdx.x;
  if (idx >= n || tx_small_job(blocks[idx])) return;
  residual_job<MODE, RDOQ ? 1024 : 4>(s, idx, orig, pred, rec, blocks, levels, level_off, nnz_out,
                                      tx_tables, lay, &rq, rq_ctx, rq_prm, dist_out);
}

#endif  // XVCGPU_K_TX_H_
