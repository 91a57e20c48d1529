// k_tx2.h -- X1, Q (QuantFast), Q1, X2, R1 for blocks up to 16x16, throughput
// form: ONE WAVEFRONT per TransformAndReconstruct call
// (transform_encoder.cc:203-285); a 1080p picture has 24480 of them (8160
// luma 16x16 + 16320 chroma 8x8), all independent.
//
// Every 1-D pass of the reference is `out = (M x in + add) >> shift` read as a
// transposing matrix product.  Here all four passes have
// the same "NT" shape  out[a][b] = sum_j A[a][j] * B[b][j]  with both operands
// contiguous in j, so a lane loads one row of A and OPL rows of B as 16-byte
// vectors and accumulates with v_dot2_i32_i16 (two int16 MACs, int32 wrap-
// around = the reference's int32 accumulation):
//   fwd 1  T[k][y]   = sum_j Mh[k][j]  * R[y][j]     (R = orig - pred)
//   fwd 2  C[x][k2]  = sum_j T[x][j]   * Mv[k2][j]   (coefficients, transposed)
//   inv 1  U[r][x]   = sum_j MvT[r][j] * C[x][j]     (after quant + dequant)
//   inv 2  res[y][c] = sum_j U[y][j]   * MhT[c][j]   -> AddClip -> rec
// Matrices (both orientations) are read from the device tables through L1;
// the three 512-byte block buffers per wave live in LDS.  No workgroup
// barrier: jobs never share data across waves.
#ifndef XVCGPU_K_TX2_H_
#define XVCGPU_K_TX2_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_me.h"
#include "k_me2.h"
#include "k_tx.h"
#include "xvcgpu_internal.h"

#define TX2_WAVES 4

typedef short tx2_v2s __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) Tx2Shared {
  int16_t r[256];  // R, later U
  int16_t t[256];  // T
  int16_t c[256];  // C (transposed coefficients / levels / dequantised)
};

__device__ __forceinline__ int tx2_dot2(uint32_t a, uint32_t b, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(tx2_v2s, a),
                                __builtin_bit_cast(tx2_v2s, b), acc, false);
}

// Load a row of NJ int16 (NJ in {4,8,16}) as NJ/2 packed dwords.
template <int NJ>
__device__ __forceinline__ void tx2_load_row(const int16_t *p, uint32_t *d) {
  if (NJ == 4) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    d[0] = v.x; d[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < NJ / 8; q++) {
      const uint4 v = *reinterpret_cast<const uint4 *>(p + 8 * q);
      d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
    }
  }
}

// out[a][b] = f((sum_j A[a][j]*B[b][j] + add) >> shift), a < na, b < nb;
// rows of A / B / out are contiguous (strides NJ, NJ, nb).  OPL outputs/lane.
template <int NJ, int OPL, bool CLIP, int G = 64>
__device__ __forceinline__ void tx2_stage(const int16_t *A, const int16_t *B,
                                          int na, int nb, int add, int shift,
                                          int16_t *out) {
  const int lane = ME2_LANE & (G - 1);
  const int gb = nb / OPL;              // a power of two (block sides are)
  const int a = lane >> (31 - __clz(gb)), b0 = (lane & (gb - 1)) * OPL;
  if (a >= na) return;
  uint32_t ra[NJ / 2];
  tx2_load_row<NJ>(A + a * NJ, ra);
  int16_t res[OPL];
#pragma unroll
  for (int o = 0; o < OPL; o++) {
    uint32_t rb[NJ / 2];
    tx2_load_row<NJ>(B + (b0 + o) * NJ, rb);
    int acc = add;
#pragma unroll
    for (int j = 0; j < NJ / 2; j++) acc = tx2_dot2(ra[j], rb[j], acc);
    int v = acc >> shift;
    if (CLIP) v = d_clip3(v, -32768, 32767);
    res[o] = (int16_t)v;
  }
  int16_t *dst = out + a * nb + b0;
  if (OPL == 4) {
    *reinterpret_cast<uint2 *>(dst) =
        make_uint2((uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16),
                   (uint16_t)res[2 % OPL] | ((uint32_t)(uint16_t)res[3 % OPL] << 16));
  } else if (OPL == 2) {
    *reinterpret_cast<uint32_t *>(dst) =
        (uint16_t)res[0] | ((uint32_t)(uint16_t)res[1 % OPL] << 16);
  } else {
    dst[0] = res[0];
  }
}

// G = lanes that work on one block: 64 (a wave per block) or 32 (two blocks of at
// most 8x8 side by side in one wave, e.g. the U and V blocks of a CU).
template <bool CLIP, int G = 64>
__device__ __forceinline__ void tx2_stage_dispatch(int nj, const int16_t *A,
                                                   const int16_t *B, int na, int nb,
                                                   int add, int shift, int16_t *out) {
  // outputs per lane so that na*nb/OPL <= G lanes
  const int total = na * nb;
  if (nj == 16) {
    if (total > 2 * G) tx2_stage<16, 4, CLIP, G>(A, B, na, nb, add, shift, out);
    else if (total > G) tx2_stage<16, 2, CLIP, G>(A, B, na, nb, add, shift, out);
    else tx2_stage<16, 1, CLIP, G>(A, B, na, nb, add, shift, out);
  } else if (nj == 8) {
    if (total > 2 * G) tx2_stage<8, 4, CLIP, G>(A, B, na, nb, add, shift, out);
    else if (total > G) tx2_stage<8, 2, CLIP, G>(A, B, na, nb, add, shift, out);
    else tx2_stage<8, 1, CLIP, G>(A, B, na, nb, add, shift, out);
  } else {
    if (total > 2 * G) tx2_stage<4, 4, CLIP, G>(A, B, na, nb, add, shift, out);
    else if (total > G) tx2_stage<4, 2, CLIP, G>(A, B, na, nb, add, shift, out);
    else tx2_stage<4, 1, CLIP, G>(A, B, na, nb, add, shift, out);
  }
}

// Sum over the G lanes that share a block, returned to all of them.
template <int G>
__device__ __forceinline__ int tx2_group_add(int v) {
  if (G == 64) return wave_reduce_add_i32(v);
  v = dpp_group_sum<16>(v);
  return v + __shfl_xor(v, 16, 64);
}

// Forward-only jobs that feed the RDO quantiser can classify their block on the
// way (xvcgpu_frame_pass: saves the separate pass over all coefficients): cls =
// RdoqLists::cls, levels / nnz = the quantiser's outputs (zeros for a block in
// which nothing quantises to a level).
struct FwdClassify {
  signed char *cls;
  int16_t *levels;
  int32_t *nnz;
  // the all-zero proof on the spot (rq_prove_zero_lds, k_rdoq.h): the quantiser's
  // context snapshots and per-block parameters, and the wave's scratch (set by the
  // kernel); null: classification only
  const xvcgpu_rdoq_contexts *rq_ctx;
  const xvcgpu_rdoq_params *rq_prm;
  RqProveLds *pv;
};

// One TransformAndReconstruct job by one wave.  pred_p / pred_stride address
// the predicted block (a picture plane in global memory, or an LDS buffer when
// the caller has just motion-compensated it).
// G = 32: the two halves of the wave run two blocks (each at most 64 samples,
// same size) side by side; `b`, `bi`, `po`, `pred_p`, `pr`, `orig_pre` are then
// per-lane values of the lane's own half and `soff` places the half's working
// set inside the shared arrays.
// FW, FH > 0: the instance for that exact block size (b.w, b.h must be it): every loop bound,
// shift and dispatch below is a constant then.
template <int MODE, int G = 64, bool RDOQ = false, int FW = 0, int FH = 0>
__device__ __forceinline__ int tx2_job(Tx2Shared &sh, const xvcgpu_tx_block &b, int bi,
                                       int bd, const PlaneView &po, const uint16_t *pred_p,
                                       int pred_stride, const PlaneView &pr,
                                       int16_t *levels, const uint32_t *level_off,
                                       int32_t *nnz_out, const int16_t *tx_tables,
                                       const int16_t *tx_tables_t,
                                       const TxTableLayout &lay,
                                       const U16x4 *orig_pre = nullptr, int soff = 0,
                                       RdoqShared<(G == 32 ? 64 : 256)> *rq = nullptr,
                                       const xvcgpu_rdoq_contexts *rq_ctx = nullptr,
                                       const xvcgpu_rdoq_params *rq_prm = nullptr,
                                       unsigned long long *dist_out = nullptr,
                                       const FwdClassify *fc = nullptr) {
  struct { int16_t *r, *t, *c; } s = {sh.r + soff, sh.t + soff, sh.c + soff};
  // dist_out: SSD between the original residual (orig - pred) and the
  // reconstructed one, >> 2 (bd - 8): SampleMetric::CompareShort on
  // temp_resi_orig_ / temp_resi_ (transform_encoder.cc:75-76, sample_metric.cc:
  // 286-290) - the RD loop's distortion of a coded inter block
  unsigned long long dist_acc = 0;
  auto dist4 = [&](int y, int x, const U16x4 &p, int r0, int r1, int r2, int r3) {
    const U16x4 o = *reinterpret_cast<const U16x4 *>(po.p + (ptrdiff_t)(b.y + y) * po.stride +
                                                     b.x + x);
    const int d0 = (int)(o.v[0] & 0xffff) - (int)(p.v[0] & 0xffff) - r0;
    const int d1 = (int)(o.v[0] >> 16) - (int)(p.v[0] >> 16) - r1;
    const int d2 = (int)(o.v[1] & 0xffff) - (int)(p.v[1] & 0xffff) - r2;
    const int d3 = (int)(o.v[1] >> 16) - (int)(p.v[1] >> 16) - r3;
    dist_acc += (unsigned long long)((long long)d0 * d0) + (unsigned long long)((long long)d1 * d1) +
                (unsigned long long)((long long)d2 * d2) + (unsigned long long)((long long)d3 * d3);
  };
  auto dist_finish = [&]() {
    unsigned long long v = dist_acc;
#pragma unroll
    for (int sft = 1; sft < G; sft <<= 1) v += __shfl_xor(v, sft, 64);
    if ((ME2_LANE & (G - 1)) == 0) dist_out[bi] = v >> (2 * (bd - 8));
  };
  const int lane = ME2_LANE & (G - 1);
  const int w = FW ? FW : b.w, h = FH ? FH : b.h;
  const int lw = 31 - __clz(w);
  const int lgw = lw, lgh = 31 - __clz(h);   // sides 4 ... 16: powers of two
  int16_t *lv = (levels && level_off) ? levels + level_off[bi] : nullptr;
  const int offh = tx_table_off(lay, b.tx_hor, w), offv = tx_table_off(lay, b.tx_ver, h);
  const int16_t *Mh = tx_tables + offh, *Mv = tx_tables + offv;
  const int16_t *MhT = tx_tables_t + offh, *MvT = tx_tables_t + offv;

  int qpb = b.qp + 6 * (bd - 8);
  qpb = qpb > 0 ? qpb : 0;
  const bool bias = ((lgw + lgh) & 1) != 0;
  const int tshift = 15 - bd - ((lgw + lgh) >> 1);
  const int n_el = w * h;

  int nnz;
  if (MODE != TX_MODE_INV) {
    // residual, 4 samples per lane along a row
    for (int i = lane * 4; i < n_el; i += 4 * G) {
      const int y = i >> lw, x = i & (w - 1);
      // blocks are <= 256 samples: one iteration, so a caller may have
      // fetched this lane's four original samples ahead of time
      const U16x4 o = orig_pre ? *orig_pre
                               : *reinterpret_cast<const U16x4 *>(
                                     po.p + (ptrdiff_t)(b.y + y) * po.stride + b.x + x);
      const U16x4 p = *reinterpret_cast<const U16x4 *>(
          pred_p + (ptrdiff_t)y * pred_stride + x);
      const int d0 = (int)(o.v[0] & 0xffff) - (int)(p.v[0] & 0xffff);
      const int d1 = (int)(o.v[0] >> 16) - (int)(p.v[0] >> 16);
      const int d2 = (int)(o.v[1] & 0xffff) - (int)(p.v[1] & 0xffff);
      const int d3 = (int)(o.v[1] >> 16) - (int)(p.v[1] >> 16);
      *reinterpret_cast<uint2 *>(s.r + i) =
          make_uint2((uint32_t)(d0 & 0xffff) | ((uint32_t)d1 << 16),
                     (uint32_t)(d2 & 0xffff) | ((uint32_t)d3 << 16));
    }
    wave_sync();
    ME2_TRACE(2);
    const int shift1 = lgw + bd - 9 + (b.tx_hor == XVC_TX_DCT2_LOW ? 0 : 2);
    const int shift2 = lgh + 6 + (b.tx_ver == XVC_TX_DCT2_LOW ? 0 : 2);
    // fwd 1: T[k][y] (w rows of h), fwd 2: C[x][k2] (w rows of h)
    tx2_stage_dispatch<false, G>(w, Mh, s.r, w, h, 1 << (shift1 - 1), shift1, s.t);
    wave_sync();
    ME2_TRACE(3);
    tx2_stage_dispatch<false, G>(h, s.t, Mv, w, h, 1 << (shift2 - 1), shift2, s.c);
    wave_sync();
    ME2_TRACE(4);
    if (MODE == TX_MODE_FWD) {
      if (fc && fc->cls) {
        // the classification pass of the RDO quantiser (rdoq_classify_kernel, k_rdoq.h)
        // on the coefficients at hand: does any of them quantise to a level at all?
        const int fq_shift = 14 + qpb / 6 + tshift + (bias ? 7 : 0);
        const int fq_scale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
        const long long fq_offset = 1ll << (fq_shift - 1);
        bool any = false;
        RqProveLds *pv = fc->pv;
        const int grp = ME2_LANE / G;
        if (pv) {
          if (lane == 0) {
            pv->n[grp] = 0;
            pv->fail[grp] = 0;
          }
          wave_sync();
        }
        for (int i = lane; i < n_el; i += G) {
          const int a = (short)d_abs((int)s.c[i]);
          const bool nz = (short)(int)((((long long)a * fq_scale) + fq_offset) >> fq_shift) != 0;
          any |= nz;
          if (pv && nz) {   // a candidate of the all-zero proof: C[x][y] at s.c[x * h + y]
            const int slot = atomicAdd(&pv->n[grp], 1);
            if (slot < RQ_PROVE_MAX_CANDS)
              pv->xy[grp][slot] = (unsigned short)(((i & (h - 1)) << 8) | (i >> lgh));
            if (a < 0) pv->fail[grp] = 1;   // a magnitude of 32768: no proof
          }
        }
        const unsigned long long group =
            G == 64 ? ~0ull : (((1ull << G) - 1) << (ME2_LANE & ~(G - 1)));
        bool live = (__ballot(any) & group) != 0;
        if (pv) {
          wave_sync();
          if (rq_prove_zero_lds<G>(*pv, b, bd, fc->rq_ctx, fc->rq_prm[bi], s.c, live)) live = false;
        }
        if (lane == 0) fc->cls[bi] = live ? (signed char)rq_class_of(b) : (signed char)-1;
        if (!live) {   // its levels are zeros; the coefficients are not needed again
          int16_t *z = fc->levels + level_off[bi];
          for (int i = lane; i < n_el; i += G) z[i] = 0;
          if (lane == 0 && fc->nnz) fc->nnz[bi] = 0;
          return 0;
        }
      }
      if (lv)
        for (int i = lane; i < n_el; i += G) {
          const int x = i >> lgh, k2 = i & (h - 1);  // C[x][k2] (h a power of two)
          lv[k2 * w + x] = s.c[i];
        }
      return 0;
    }
    // QuantFast (rdo_quant.cc:156-201): levels -> s.r, rounding remainders
    // -> s.t, the coefficients stay in s.c (same [x][k2] layout)
    const bool intra_pic = (b.intra_pic & XVC_TXF_INTRA_PIC) != 0;
    const bool sign_hide = !(b.intra_pic & XVC_TXF_NO_SIGN_HIDING);
    const int scan_order = (b.intra_pic >> XVC_TXF_SCAN_SHIFT) & 3;
    const int qshift = 14 + qpb / 6 + tshift + (bias ? 7 : 0);
    const int qscale = kFwdQuantScales[qpb % 6] * (bias ? 181 : 1);
    const long long qoff = (long long)((intra_pic ? 171ull : 85ull) << (qshift - 9));
    // (with G = 32 both halves of the wave take the same branch: the caller
    // gives the two blocks the same flags)
    const bool use_rdoq = RDOQ && (b.intra_pic & XVC_TXF_RDOQ) != 0;
    if (RDOQ && use_rdoq) {
      // RdoQuant::QuantRdo (rdo_quant.cc:203-446): levels -> s.r, same layout
      const xvcgpu_rdoq_params prm = rq_prm[bi];
      const int16_t *cfp = s.c;
      int16_t *lvp = s.r;
      auto cf_at = [cfp, h](int x, int y) { return (int)cfp[x * h + y]; };
      auto lv_at = [lvp, h](int x, int y) { return lvp + x * h + y; };
      if (G == 64 && rq4_takes(64, w, h, scan_order)) {
        // four lanes per sub-block (k_rdoq4.h): the whole wave on this block's walk
        rq_stage_costs(&rq_ctx[prm.ctx_index], rq->ctx_bits, lane, 64);
        wave_sync();
        nnz = wave_rdoq4<64, 1>(*rq, lane, bd, w, h, b.qp, b.comp == 0, sign_hide, prm, cf_at,
                                lv_at);
      } else {
        nnz = wave_rdoq<G>(*rq, lane, bd, w, h, b.qp, b.comp == 0, scan_order, sign_hide,
                           rq_ctx[prm.ctx_index], prm, cf_at, lv_at);
      }
      wave_sync();
    }
    int local = 0;
    if (!use_rdoq) {
    for (int i = lane; i < n_el; i += G) {
      const int v = s.c[i];
      const int sign = v < 0 ? -1 : 1;
      const long long abs_coeff = d_abs(v);
      const int level = (int)(((abs_coeff * qscale) + qoff) >> qshift);
      local += level != 0;
      s.r[i] = (int16_t)d_clip3(level * sign, -32768, 32767);
      s.t[i] = (int16_t)(((abs_coeff * qscale) - ((long long)level << qshift)) >> (qshift - 8));
    }
    nnz = tx2_group_add<G>(local);
    }
    // CoeffSignHideFast (rdo_quant.cc:448-573): lane = 4x4 sub-block (<= 16)
    if (!use_rdoq && sign_hide && nnz > 1 && w >= 4 && h >= 4) {
      wave_sync();
      const int gw = w >> 2, gh = h >> 2;
      auto idx = [h](int x, int y) { return x * h + y; };
      const bool mine = lane < gw * gh;
      const int sx = mine ? (lane & (gw - 1)) : 0, sy = mine ? (lane >> (lgw - 2)) : 0;
      bool any = false;
      if (mine)
        for (int k = 0; k < 16; k++) any |= s.r[idx(4 * sx + (k & 3), 4 * sy + (k >> 2))] != 0;
      const int my_scan = d_sb_scan_index(scan_order, gw, gh, sx, sy);
      // highest scan index among the sub-blocks that hold a level
      int last_sb = any ? my_scan : -1;
#pragma unroll
      for (int sft = 1; sft < 16; sft <<= 1) {
        const int o = __shfl_xor(last_sb, sft, 64);
        last_sb = o > last_sb ? o : last_sb;
      }
      int dn = 0;
      if (mine)
        dn = d_sign_hide_subblock(scan_order, 4 * sx, 4 * sy, my_scan == last_sb, s.r, s.t,
                                  s.c, idx);
      nnz += tx2_group_add<G>(dn);
    }
    if (nnz_out && lane == 0) nnz_out[bi] = nnz;
    if (lv) {
      wave_sync();
      for (int i = lane; i < n_el; i += G) {
        const int x = i >> lgh, k2 = i & (h - 1);
        lv[k2 * w + x] = s.r[i];
      }
    }
  } else {
    nnz = nnz_out[bi];
    if (nnz)
      for (int i = lane; i < n_el; i += G) {
        const int x = i >> lgh, k2 = i & (h - 1);
        s.r[i] = lv[k2 * w + x];
      }
  }
  wave_sync();

  if (nnz == 0) {  // cbf == 0: rec = pred
    // in place (the prediction was written into the reconstruction's picture,
    // xvcgpu_inv_transform_batch with pred == rec): nothing to do - at QP 32 that
    // is 7 of 8 blocks of a picture
    if (MODE == TX_MODE_INV && !dist_out &&
        pred_p == pr.p + (ptrdiff_t)b.y * pr.stride + b.x)
      return 0;
    for (int i = lane * 4; i < n_el; i += 4 * G) {
      const int y = i >> lw, x = i & (w - 1);
      const U16x4 p = *reinterpret_cast<const U16x4 *>(
          pred_p + (ptrdiff_t)y * pred_stride + x);
      *reinterpret_cast<U16x4 *>(pr.p + (ptrdiff_t)(b.y + y) * pr.stride + b.x + x) = p;
      if (dist_out) dist4(y, x, p, 0, 0, 0, 0);
    }
    if (dist_out) dist_finish();
    return 0;
  }
  const bool dc_only = nnz == 1 && s.r[0] != 0;
  wave_sync();
  ME2_TRACE(5);
  // Quantize::Inverse (quantize.cc:94-125): levels in s.r -> s.c
  {
    const int shift = 6 - tshift + (bias ? 8 : 0);
    const int scale = (kInvQuantScales[qpb % 6] << (qpb / 6)) * (bias ? 181 : 1);
    for (int i = lane; i < n_el; i += G) {
      const int prod = (int)s.r[i] * scale;
      int cf;
      if (shift > 0) cf = (prod + (1 << (shift - 1))) >> shift;
      else cf = (int)((unsigned)prod << -shift);
      s.c[i] = (int16_t)d_clip3(cf, -32768, 32767);
    }
  }
  wave_sync();
  ME2_TRACE(6);
  const int smax = (1 << bd) - 1;
  const bool dct2_both =
      (b.tx_ver == XVC_TX_DEFAULT || b.tx_ver == XVC_TX_DCT2 || b.tx_ver == XVC_TX_DCT2_LOW) &&
      (b.tx_hor == XVC_TX_DEFAULT || b.tx_hor == XVC_TX_DCT2 || b.tx_hor == XVC_TX_DCT2_LOW);
  if (dc_only && dct2_both) {  // InvDct2Dc, transform.cc:279-291
    const int sh = 14 - bd, add = 1 << (sh - 1);
    const int cf = (int16_t)(((((int)s.c[0] + 1) >> 1) + add) >> sh);
    for (int i = lane * 4; i < n_el; i += 4 * G) {
      const int y = i >> lw, x = i & (w - 1);
      const U16x4 p = *reinterpret_cast<const U16x4 *>(
          pred_p + (ptrdiff_t)y * pred_stride + x);
      U16x4 o;
      o.v[0] = (uint32_t)d_clip3((int)(p.v[0] & 0xffff) + cf, 0, smax) |
               ((uint32_t)d_clip3((int)(p.v[0] >> 16) + cf, 0, smax) << 16);
      o.v[1] = (uint32_t)d_clip3((int)(p.v[1] & 0xffff) + cf, 0, smax) |
               ((uint32_t)d_clip3((int)(p.v[1] >> 16) + cf, 0, smax) << 16);
      *reinterpret_cast<U16x4 *>(pr.p + (ptrdiff_t)(b.y + y) * pr.stride + b.x + x) = o;
      if (dist_out) dist4(y, x, p, cf, cf, cf, cf);
    }
    if (dist_out) dist_finish();
    return nnz;
  }
  // inverse: U[r][x] (h rows of w) into s.r, then residual rows into s.t
  {
    const int shift1 = 7 + (b.tx_ver == XVC_TX_DCT2_LOW ? 0 : 2);
    const int shift2 = 20 - bd + (b.tx_hor == XVC_TX_DCT2_LOW ? 0 : 2);
    tx2_stage_dispatch<true, G>(h, MvT, s.c, h, w, 1 << (shift1 - 1), shift1, s.r);
    wave_sync();
    tx2_stage_dispatch<true, G>(w, s.r, MhT, h, w, 1 << (shift2 - 1), shift2, s.t);
    wave_sync();
    ME2_TRACE(7);
  }
  // SampleBuffer::AddClip
  for (int i = lane * 4; i < n_el; i += 4 * G) {
    const int y = i >> lw, x = i & (w - 1);
    const U16x4 p = *reinterpret_cast<const U16x4 *>(
        pred_p + (ptrdiff_t)y * pred_stride + x);
    const uint2 rs = *reinterpret_cast<const uint2 *>(s.t + i);
    U16x4 o;
    o.v[0] = (uint32_t)d_clip3((int)(p.v[0] & 0xffff) + (int)(int16_t)(rs.x & 0xffff), 0, smax) |
             ((uint32_t)d_clip3((int)(p.v[0] >> 16) + ((int)rs.x >> 16), 0, smax) << 16);
    o.v[1] = (uint32_t)d_clip3((int)(p.v[1] & 0xffff) + (int)(int16_t)(rs.y & 0xffff), 0, smax) |
             ((uint32_t)d_clip3((int)(p.v[1] >> 16) + ((int)rs.y >> 16), 0, smax) << 16);
    *reinterpret_cast<U16x4 *>(pr.p + (ptrdiff_t)(b.y + y) * pr.stride + b.x + x) = o;
    if (dist_out)
      dist4(y, x, p, (int)(int16_t)(rs.x & 0xffff), (int)rs.x >> 16, (int)(int16_t)(rs.y & 0xffff),
            (int)rs.y >> 16);
  }
  if (dist_out) dist_finish();
  return nnz;
}

// grid: XCD-swizzled workgroups of TX2_WAVES waves; one job per wave.
// tx_tables_t: the same matrices transposed (same layout offsets).
template <int MODE, bool RDOQ>
__device__ __forceinline__ void residual_wave_kernel_body(PicView orig, PicView pred, PicView rec, const xvcgpu_tx_block *blocks, int n, int16_t *levels, const uint32_t *level_off, int32_t *nnz_out, const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, unsigned long long *dist_out) {
  __shared__ Tx2Shared s_all[TX2_WAVES];
  __shared__ RdoqShared<256> rq_all[RDOQ ? TX2_WAVES : 1];
  Tx2Shared &s = s_all[threadIdx.x >> 6];
  const int n_wg = (n + TX2_WAVES - 1) / TX2_WAVES;
  const int wg = xcd_job_index(blockIdx.x, n_wg);
  if (wg < 0) return;
  const int bi = __builtin_amdgcn_readfirstlane(wg * TX2_WAVES + (int)(threadIdx.x >> 6));
  if (bi >= n) return;
  // in place and nothing coded: the block is already what it will be
  if (MODE == TX_MODE_INV && !dist_out && pred.c[0].p == rec.c[0].p && nnz_out[bi] == 0) return;
  const xvcgpu_tx_block b = blocks[bi];
  if (!tx_small_job(b)) return;  // general path: residual_kernel<>
  const PlaneView pp = pred.c[b.comp];
  tx2_job<MODE, 64, RDOQ>(s, b, bi, pred.bd, orig.c[b.comp],
                          pp.p + (ptrdiff_t)b.y * pp.stride + b.x, pp.stride, rec.c[b.comp],
                          levels, level_off, nnz_out, tx_tables, tx_tables_t, lay, nullptr, 0,
                          &rq_all[RDOQ ? (threadIdx.x >> 6) : 0], rq_ctx, rq_prm, dist_out);
}

// The inverse half of a frame pass whose transform blocks come in CU order, Y U V
// each (block 3 * cu + comp, CUs up to 16x16), in place on `rec` (which holds the
// prediction): two waves per CU as in recon_from_me_kernel - one for the luma
// block, one for the U and V blocks side by side, 32 lanes each.  One wave per
// block is 24 480 waves for a 1080p picture, two thirds of them for 8x8 blocks
// that leave three quarters of their wave idle; a wave whose blocks carry no
// level retires after a look at their counts.  grid: XCD-swizzled workgroups of
// TX2_WAVES waves over 2 * n_cus jobs.
__global__ void __launch_bounds__(64 * TX2_WAVES)
inv_cu_pairs_kernel(PicView rec, const xvcgpu_tx_block *blocks, int n_cus, int16_t *levels,
                    const uint32_t *level_off, int32_t *nnz_out, const int16_t *tx_tables,
                    const int16_t *tx_tables_t, TxTableLayout lay) {
  __shared__ Tx2Shared s_all[TX2_WAVES];
  Tx2Shared &s = s_all[threadIdx.x >> 6];
  const int n = 2 * n_cus;
  const int n_wg = (n + TX2_WAVES - 1) / TX2_WAVES;
  const int wg = xcd_job_index(blockIdx.x, n_wg);
  if (wg < 0) return;
  const int job = __builtin_amdgcn_readfirstlane(wg * TX2_WAVES + (int)(threadIdx.x >> 6));
  if (job >= n) return;
  const int ci = job >> 1;
  if (!(job & 1)) {
    const int bi = 3 * ci;
    if (nnz_out[bi] == 0) return;   // in place: the block is already what it will be
    const xvcgpu_tx_block b = blocks[bi];
    if (!tx_small_job(b)) return;   // (not a block of this layout: left as it is)
    const PlaneView pc = rec.c[0];
    tx2_job<TX_MODE_INV, 64, false>(s, b, bi, rec.bd, pc, pc.p + (ptrdiff_t)b.y * pc.stride + b.x,
                                    pc.stride, pc, levels, level_off, nnz_out, tx_tables,
                                    tx_tables_t, lay);
    return;
  }
  if (nnz_out[3 * ci + 1] == 0 && nnz_out[3 * ci + 2] == 0) return;
  // this lane's half: lanes 0-31 the U block, 32-63 the V block
  const int g = ME2_LANE >> 5, bi = 3 * ci + 1 + g;
  const xvcgpu_tx_block b = blocks[bi];
  if (!tx_small_job(b) || b.w * b.h > 64) return;   // a half-wave holds a block of <= 64 samples
  const PlaneView pc = g ? rec.c[2] : rec.c[1];
  tx2_job<TX_MODE_INV, 32, false>(s, b, bi, rec.bd, pc, pc.p + (ptrdiff_t)b.y * pc.stride + b.x,
                                  pc.stride, pc, levels, level_off, nnz_out, tx_tables,
                                  tx_tables_t, lay, nullptr, g * 128);
}

template <int MODE, bool RDOQ = false>
__global__ void __launch_bounds__(64 * TX2_WAVES)
residual_wave_kernel(PicView orig, PicView pred, PicView rec, const xvcgpu_tx_block *blocks, int n, int16_t *levels, const uint32_t *level_off, int32_t *nnz_out, const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx = nullptr, const xvcgpu_rdoq_params *rq_prm = nullptr, unsigned long long *dist_out = nullptr) {
  residual_wave_kernel_body<MODE, RDOQ>(orig, pred, rec, blocks, n, levels, level_off, nnz_out, tx_tables, tx_tables_t, lay, rq_ctx, rq_prm, dist_out);
}

// The transform blocks of ONE CU state's evaluation (a handful, small and large mixed)
// in one launch: a workgroup per block, which takes the one-wave path (tx2_job) or the
// workgroup path (residual_job) as residual_wave_kernel / residual_per_job_kernel
// would - the same functions, so the same results.  The two kernels one after the
// other are two wave lives (16 + 25 us with RDOQ) for a batch that fills neither.
// grid: n; block: TX_THREADS.
template <int MODE, bool RDOQ>
__device__ __forceinline__ void
residual_cu_body(const PicView &orig, const PicView &pred, const PicView &rec, const xvcgpu_tx_block *blocks, int n,
                   int16_t *levels, const uint32_t *level_off, int32_t *nnz_out,
                   const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay,
                   const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm,
                   const xvcgpu_block_pos *src_pos, const xvcgpu_eval_cand *ecands, int n_head,
                   uint64_t *eout, int strength) {
  // src_pos (xvcgpu_residual_rdoq_batch_at): block i reads its original at src_pos[2 i]
  // of `orig` and its prediction at src_pos[2 i + 1] of `pred` (positions in the plane of
  // the block's component); the reconstruction goes to the block's own (x, y) of `rec`.
  // ecands: the evaluation's distortions in the same launch (xvcgpu_eval_dist_batch's
  // arithmetic): candidates [0, n_head) - prediction against original - by workgroups
  // n .. n + n_head - 1, candidate n_head + i by the workgroup that reconstructed block i.
  struct Big {
    TxShared s;
    RdoqShared<RDOQ ? 1024 : 4> rq;
  };
  struct Small {
    Tx2Shared s;
    RdoqShared<256> rq;
  };
  constexpr size_t kBytes = sizeof(Big) > sizeof(Small) ? sizeof(Big) : sizeof(Small);
  __shared__ __attribute__((aligned(16))) unsigned char raw[kBytes];
  const int idx = blockIdx.x;
  auto price = [&](int c) {   // (one wave)
    const xvcgpu_eval_cand cd = ecands[c];
    const PlaneView pa = orig.c[cd.comp], pb = cd.versus ? rec.c[cd.comp] : pred.c[cd.comp];
    const uint16_t *a = cd.orig_at ? pa.p + (ptrdiff_t)cd.oy * pa.stride + cd.ox
                                   : pa.p + (ptrdiff_t)cd.y * pa.stride + cd.x;
    const uint16_t *bq = pb.p + (ptrdiff_t)cd.y * pb.stride + cd.x;
    const uint64_t dist = wave_compare(cd.metric, orig.bd, cd.qp, strength, cd.w, cd.h, a,
                                       pa.stride, bq, pb.stride);
    if ((threadIdx.x & 63) == 0) eout[c] = (uint64_t)((double)dist * cd.weight);
  };
  if (idx >= n) {
    if (ecands && idx - n < n_head && threadIdx.x < 64) price(idx - n);
    return;
  }
  // in place and nothing coded: the block is already what it will be
  if (MODE == TX_MODE_INV && pred.c[0].p == rec.c[0].p && nnz_out[idx] == 0) return;
  const xvcgpu_tx_block b = blocks[idx];
  const PlaneView pp = pred.c[b.comp];
  // the original as if it lay at the block's own position, the prediction by pointer
  PicView orig_v = orig;
  const uint16_t *pred_p = pp.p + (ptrdiff_t)b.y * pp.stride + b.x;
  if (src_pos) {
    const xvcgpu_block_pos so = src_pos[2 * idx], sp = src_pos[2 * idx + 1];
    PlaneView &po = orig_v.c[b.comp];
    po.p += (ptrdiff_t)(so.y - b.y) * po.stride + (so.x - b.x);
    pred_p = pp.p + (ptrdiff_t)sp.y * pp.stride + sp.x;
  }
  if (tx_small_job(b)) {
    // the first wave's job; the others wait at the barrier below (when there is one) - no
    // wave leaves in front of a barrier the rest of its workgroup still reaches
    if (!ecands && threadIdx.x >= 64) return;
    if (threadIdx.x < 64) {
      Small &u = *reinterpret_cast<Small *>(raw);
      tx2_job<MODE, 64, RDOQ>(u.s, b, idx, pred.bd, orig_v.c[b.comp], pred_p, pp.stride,
                              rec.c[b.comp], levels, level_off, nnz_out, tx_tables, tx_tables_t,
                              lay, nullptr, 0, &u.rq, rq_ctx, rq_prm, nullptr);
    }
  } else {
    Big &u = *reinterpret_cast<Big *>(raw);
    residual_job<MODE, RDOQ ? 1024 : 4>(u.s, idx, orig_v, pred, rec, blocks, levels, level_off,
                                        nnz_out, tx_tables, lay, &u.rq, rq_ctx, rq_prm, nullptr,
                                        src_pos ? pred_p : nullptr, pp.stride);
  }
  if (ecands) {
    // the block's reconstruction was written by this workgroup: visible to its first
    // wave behind a workgroup-scope release / acquire around the barrier (every wave of
    // the workgroup arrives here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x < 64) price(n_head + idx);
  }
}

template <int MODE, bool RDOQ>
__global__ void __launch_bounds__(TX_THREADS)
residual_cu_kernel(PicView orig, PicView pred, PicView rec, const xvcgpu_tx_block *blocks, int n,
                   int16_t *levels, const uint32_t *level_off, int32_t *nnz_out,
                   const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay,
                   const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm,
                   const xvcgpu_block_pos *src_pos = nullptr,
                   const xvcgpu_eval_cand *ecands = nullptr, int n_head = 0,
                   uint64_t *eout = nullptr, int strength = 0) {
  residual_cu_body<MODE, RDOQ>(orig, pred, rec, blocks, n, levels, level_off, nnz_out, tx_tables,
                               tx_tables_t, lay, rq_ctx, rq_prm, src_pos, ecands, n_head, eout,
                               strength);
}

#endif  // XVCGPU_K_TX2_H_
