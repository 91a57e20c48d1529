// k_recon.h -- I1 + X1 + Q + Q1 + X2 + R1 fused per (CU, component): what
// InterSearch::CompressAndEvalCbf does for a uni-pred CU once the MV is known
// (inter_search.cc:261-365: MotionCompensation, then
// TransformEncoder::TransformAndReconstruct, transform_encoder.cc:203-285),
// with the MV taken from the motion-search result still resident in HBM.
// One wave per (CU, component); the prediction never leaves LDS, and the luma
// wave also stores the CU's deblocking metadata (what CuEncoder writes into
// CodingUnit, cu_encoder.cc:543-577).  CUs up to 16x16 (chroma >= 4x4).
#ifndef XVCGPU_K_RECON_H_
#define XVCGPU_K_RECON_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_me2.h"
#include "k_tx2.h"
#include "xvcgpu_internal.h"

struct __attribute__((aligned(16))) ReconShared {
  Tx2Shared tx;
  uint16_t pred[256];
  int16_t tmp[16 * 23];
  uint16_t win[23 * 24];  // reference window of the block, rows/cols -3..+4
};

// MotionCompUniPred -> Sample by one wave (same arithmetic as wg_interp_block
// in k_interp.h; inter_prediction.cc:1138-1154, :1207-1448).
template <bool CHROMA>
__device__ __forceinline__ void wave_interp_block(int bd, int w, int h, int fx,
                                                  int fy, const uint16_t *ref, int rs,
                                                  int16_t *tmp, uint16_t *dst) {
  constexpr int N = CHROMA ? 4 : 8;
  constexpr int BACK = N / 2 - 1;
  const int lane = ME2_LANE;
  const int smax = (1 << bd) - 1;
  const int lw = 31 - __clz(w);
  const int16_t *fh = CHROMA ? kChromaTaps[fx] : kLumaTaps[fx];
  const int16_t *fv = CHROMA ? kChromaTaps[fy] : kLumaTaps[fy];
  if (fx == 0 && fy == 0) {
    for (int i = lane; i < w * h; i += 64)
      dst[i] = ref[(ptrdiff_t)(i >> lw) * rs + (i & (w - 1))];
    return;
  }
  if (fy == 0) {
    for (int i = lane; i < w * h; i += 64) {
      const uint16_t *s = ref + (ptrdiff_t)(i >> lw) * rs + (i & (w - 1)) - BACK;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k] * fh[k];
      dst[i] = d_clip_bd((sum + 32) >> 6, smax);
    }
    return;
  }
  if (fx == 0) {
    for (int i = lane; i < w * h; i += 64) {
      const uint16_t *s = ref + (ptrdiff_t)((i >> lw) - BACK) * rs + (i & (w - 1));
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[(ptrdiff_t)k * rs] * fv[k];
      dst[i] = d_clip_bd((int16_t)((sum + 32) >> 6), smax);
    }
    return;
  }
  {
    const int shift = 6 - (14 - bd), offset = -(8192 << shift);
    for (int i = lane; i < w * (h + N - 1); i += 64) {
      const uint16_t *s =
          ref + (ptrdiff_t)((i >> lw) - BACK) * rs + (i & (w - 1)) - BACK;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k] * fh[k];
      tmp[i] = (int16_t)((sum + offset) >> shift);
    }
  }
  wave_sync();
  {
    const int shift = 6 + (14 - bd);
    const int offset = (8192 << 6) + (1 << (shift - 1));
    for (int i = lane; i < w * h; i += 64) {
      const int16_t *s = tmp + i;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k * w] * fv[k];
      dst[i] = d_clip_bd((int16_t)((sum + offset) >> shift), smax);
    }
  }
}

// Same arithmetic with the reference window staged in LDS first: one batch of
// 16-byte loads (one memory round trip) instead of a dependent load per filter
// tap and loop iteration, and the horizontal pass on packed pairs
// (v_dot2c_i32_i16: even outputs use the tap pairs as they are, odd outputs
// the set shifted by one sample).  Blocks up to 16x16.
template <bool CHROMA, int FW = 0, int FH = 0>   // FW, FH > 0: that exact block size
__device__ __forceinline__ void wave_interp_block_lds(int bd, int w_in, int h_in, int fx, int fy,
                                                      const uint16_t *ref, int rs,
                                                      uint16_t *win, int16_t *tmp,
                                                      uint16_t *dst) {
  constexpr int N = CHROMA ? 4 : 8;
  constexpr int BACK = N / 2 - 1;
  constexpr int NP = N / 2;  // tap pairs
  const int w = FW ? FW : w_in, h = FH ? FH : h_in;
  const int lane = ME2_LANE;
  const int smax = (1 << bd) - 1;
  const int lw = 31 - __clz(w);
  const int ws = (w + N - 1 + 7) & ~7;  // window row stride, 16-byte rows
  const int16_t *fh = CHROMA ? kChromaTaps[fx] : kLumaTaps[fx];
  const int16_t *fv = CHROMA ? kChromaTaps[fy] : kLumaTaps[fy];
  wave_copy_chunks(win, ws, ref - (ptrdiff_t)BACK * rs - BACK, rs, h + N - 1, ws >> 3);
  wave_sync();
  if (fx == 0 && fy == 0) {
    for (int i = lane; i < w * h; i += 64)
      dst[i] = win[((i >> lw) + BACK) * ws + (i & (w - 1)) + BACK];
    return;
  }
  if (fx == 0) {  // FilterVerSampleSample (narrows to int16 before the clip)
    for (int i = lane; i < w * h; i += 64) {
      const uint16_t *s = win + (i >> lw) * ws + (i & (w - 1)) + BACK;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k * ws] * fv[k];
      dst[i] = d_clip_bd((int16_t)((sum + 32) >> 6), smax);
    }
    return;
  }
  // horizontal pass, two outputs (x0 even, x0 + 1) per lane
  uint32_t ta[NP], tb[NP + 1];
#pragma unroll
  for (int k = 0; k < NP; k++) ta[k] = sp_pack_taps(fh[2 * k], fh[2 * k + 1]);
  tb[0] = sp_pack_taps(0, fh[0]);
#pragma unroll
  for (int k = 1; k < NP; k++) tb[k] = sp_pack_taps(fh[2 * k - 1], fh[2 * k]);
  tb[NP] = sp_pack_taps(fh[N - 1], 0);
  const uint32_t *win32 = reinterpret_cast<const uint32_t *>(win);
  const int hw = w >> 1, lhw = lw - 1;
  const bool two_stage = fy != 0;
  const int rows = two_stage ? h + N - 1 : h, r0 = two_stage ? 0 : BACK;
  const int shift = 6 - (14 - bd), offset = -(8192 << shift);
  for (int i = lane; i < rows * hw; i += 64) {
    const int r = i >> lhw, x0 = (i & (hw - 1)) << 1;
    const uint32_t *d = win32 + (((r + r0) * ws + x0) >> 1);
    uint32_t dv[NP + 1];
#pragma unroll
    for (int k = 0; k <= NP; k++) dv[k] = d[k];
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) s0 = sp_dot2(dv[k], ta[k], s0);
#pragma unroll
    for (int k = 0; k <= NP; k++) s1 = sp_dot2(dv[k], tb[k], s1);
    if (two_stage) {
      tmp[r * w + x0] = (int16_t)((s0 + offset) >> shift);
      tmp[r * w + x0 + 1] = (int16_t)((s1 + offset) >> shift);
    } else {
      dst[r * w + x0] = d_clip_bd((s0 + 32) >> 6, smax);
      dst[r * w + x0 + 1] = d_clip_bd((s1 + 32) >> 6, smax);
    }
  }
  if (!two_stage) return;
  wave_sync();
  {
    const int sh2 = 6 + (14 - bd);
    const int off2 = (8192 << 6) + (1 << (sh2 - 1));
    for (int i = lane; i < w * h; i += 64) {
      const int16_t *s = tmp + i;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k * w] * fv[k];
      dst[i] = d_clip_bd((int16_t)((sum + off2) >> sh2), smax);
    }
  }
}

// grid: XCD-swizzled workgroups of 4 waves; two waves per CU: one for the luma
// block, one for the U and V blocks side by side (32 lanes each: a chroma
// block of a CU up to 16x16 has at most 64 samples, which would leave three
// quarters of a wave of its own idle through the whole pipeline).
// RDOQ: quantise with RdoQuant::QuantRdo (k_rdoq.h); rq_prm[3 * cu + comp].
// FWD: the front half only - prediction (also written to the picture `rec`,
// which then plays TransformEncoder's prediction buffer) and forward transform,
// coefficients to coeffs + coeff_off[3 * cu + comp]: what precedes a quantiser
// that runs as its own kernel (xvcgpu_quant_rdo_batch).
template <bool RDOQ, bool FWD>
__device__ __forceinline__ void recon_from_me_kernel_body(PicView orig, PicView ref, PicView rec, const xvcgpu_me_block *blocks, const xvcgpu_me_result *results, int n_cus, int qp_y, int qp_c, int intra_pic, int ref_poc, int32_t *nnz_out, xvcgpu_cu_info *cus, const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx, const xvcgpu_rdoq_params *rq_prm, int16_t *coeffs, const uint32_t *coeff_off, FwdClassify fc) {
  constexpr int TXM = FWD ? TX_MODE_FWD : TX_MODE_FULL;
  __shared__ ReconShared s_all[4];
  // one scratch per wave; a chroma wave splits it between its two halves
  // per wave: one 16x16 luma block, or two 8x8 chroma blocks side by side
  union RqWave {
    RdoqShared<256> one;
    RdoqShared<64> two[2];
  };
  __shared__ RqWave rq_all[RDOQ ? 4 : 1];
  RdoqShared<256> *rq_wave = &rq_all[RDOQ ? (threadIdx.x >> 6) : 0].one;
  ReconShared &s = s_all[threadIdx.x >> 6];
  // the forward half's classification can prove blocks all zero on the spot: a
  // scratch per wave for it
  FwdClassify fcl = fc;
  if constexpr (FWD) {
    __shared__ RqProveLds pv_all[4];
    fcl.pv = (fc.cls && fc.rq_ctx && fc.rq_prm) ? &pv_all[threadIdx.x >> 6] : nullptr;
    if (fcl.pv && (threadIdx.x & 63) == 0) fcl.pv->staged_ctx = -1;
  }
  const int n = n_cus * 2;
  const int n_wg = (n + 3) / 4;
  const int wg = xcd_job_index(blockIdx.x, n_wg);
  if (wg < 0) return;
  const int job = __builtin_amdgcn_readfirstlane(wg * 4 + (int)(threadIdx.x >> 6));  // wave-uniform
  if (job >= n) return;
  const int ci = job >> 1;
  const bool chroma = (job & 1) != 0;
#ifdef XVCGPU_TRACE
  const int bi = job;
#endif
  ME2_TRACE(0);
  const xvcgpu_me_block mb = blocks[ci];
  const xvcgpu_me_result mr = results[ci];
  const int bd = ref.bd;
  // MotionCompensationMv: clip, split (GetFullpelRef, 4:2:0)
  int mx = mr.mv_x, my = mr.mv_y;
  d_clip_mv(mb.x, mb.y, ref.c[0].w, ref.c[0].h, mx, my);
  const int cs = chroma ? 1 : 0, shift = 4 + cs;
  const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
  const int cx = mb.x >> cs, cy = mb.y >> cs, cw = mb.w >> cs, ch = mb.h >> cs;
  xvcgpu_tx_block tb;
  tb.x = (int16_t)cx;
  tb.y = (int16_t)cy;
  tb.w = (uint8_t)cw;
  tb.h = (uint8_t)ch;
  tb.tx_hor = XVC_TX_DEFAULT;
  tb.tx_ver = XVC_TX_DEFAULT;
  tb.dst4x4 = 0;
  tb.intra_pic = (uint8_t)intra_pic;
  if (chroma) {
    // this lane's half: lanes 0-31 the U block, 32-63 the V block
    const int g = ME2_LANE >> 5, comp = 1 + g;
    const PlaneView po = g ? orig.c[2] : orig.c[1];
    const PlaneView pc = g ? rec.c[2] : rec.c[1];
    U16x4 orig_pre = {{0u, 0u}};
    {
      const int i = (ME2_LANE & 31) * 4;
      if (i < cw * ch) {
        const int lw = 31 - __clz(cw);
        orig_pre = *reinterpret_cast<const U16x4 *>(
            po.p + (ptrdiff_t)(cy + (i >> lw)) * po.stride + cx + (i & (cw - 1)));
      }
    }
    // the two predictions one after the other by the whole wave (the window
    // and the intermediate rows are reused), then both blocks together
#pragma unroll
    for (int c = 1; c <= 2; c++) {
      const PlaneView prf = ref.c[c];
      const uint16_t *r =
          prf.p + (ptrdiff_t)(cy + (my >> shift)) * prf.stride + cx + (mx >> shift);
      if (__builtin_expect(cw == 8 && ch == 8, 1))    // (the chroma blocks of a 16x16 CU: nearly every job)
        wave_interp_block_lds<true, 8, 8>(bd, cw, ch, fx, fy, r, prf.stride, s.win, s.tmp,
                                          s.pred + (c - 1) * 64);
      else
        wave_interp_block_lds<true>(bd, cw, ch, fx, fy, r, prf.stride, s.win, s.tmp,
                                    s.pred + (c - 1) * 64);
      wave_sync();
    }
    ME2_TRACE(1);
    tb.comp = (uint8_t)comp;
    tb.qp = (int8_t)qp_c;
    if (FWD) {  // the prediction, for the inverse path later
      const int i = (ME2_LANE & 31) * 4;
      if (i < cw * ch) {
        const int lw = 31 - __clz(cw);
        *reinterpret_cast<U16x4 *>(pc.p + (ptrdiff_t)(cy + (i >> lw)) * pc.stride + cx +
                                   (i & (cw - 1))) =
            *reinterpret_cast<const U16x4 *>(s.pred + g * 64 + i);
      }
    }
    if (__builtin_expect(cw == 8 && ch == 8, 1))
      tx2_job<TXM, 32, RDOQ, 8, 8>(
          s.tx, tb, 3 * ci + comp, bd, po, s.pred + g * 64, cw, pc, FWD ? coeffs : nullptr,
          FWD ? coeff_off : nullptr, nnz_out, tx_tables, tx_tables_t, lay, &orig_pre, g * 128,
          reinterpret_cast<RdoqShared<64> *>(rq_wave) + g, rq_ctx, rq_prm, nullptr,
          FWD ? &fcl : nullptr);
    else
      tx2_job<TXM, 32, RDOQ>(
          s.tx, tb, 3 * ci + comp, bd, po, s.pred + g * 64, cw, pc, FWD ? coeffs : nullptr,
          FWD ? coeff_off : nullptr, nnz_out, tx_tables, tx_tables_t, lay, &orig_pre, g * 128,
          reinterpret_cast<RdoqShared<64> *>(rq_wave) + g, rq_ctx, rq_prm, nullptr,
          FWD ? &fcl : nullptr);
    ME2_TRACE(8);
    return;
  }
  const PlaneView prf = ref.c[0];
  const uint16_t *r = prf.p + (ptrdiff_t)(cy + (my >> shift)) * prf.stride + cx + (mx >> shift);
  // this lane's four original samples for the residual: fetched now, together
  // with the reference window, instead of after the interpolation
  U16x4 orig_pre = {{0u, 0u}};
  {
    const int i = ME2_LANE * 4;
    if (i < cw * ch) {
      const int lw = 31 - __clz(cw);
      const PlaneView po = orig.c[0];
      orig_pre = *reinterpret_cast<const U16x4 *>(
          po.p + (ptrdiff_t)(cy + (i >> lw)) * po.stride + cx + (i & (cw - 1)));
    }
  }
  const bool sq16 = __builtin_expect(cw == 16 && ch == 16, 1);    // (a 16x16 CU's luma block: nearly every job)
  if (sq16) wave_interp_block_lds<false, 16, 16>(bd, cw, ch, fx, fy, r, prf.stride, s.win, s.tmp, s.pred);
  else wave_interp_block_lds<false>(bd, cw, ch, fx, fy, r, prf.stride, s.win, s.tmp, s.pred);
  wave_sync();
  ME2_TRACE(1);
  tb.comp = 0;
  tb.qp = (int8_t)qp_y;
  if (FWD) {
    const PlaneView pc = rec.c[0];
    const int lw = 31 - __clz(cw);
    for (int i = ME2_LANE * 4; i < cw * ch; i += 256)
      *reinterpret_cast<U16x4 *>(pc.p + (ptrdiff_t)(cy + (i >> lw)) * pc.stride + cx +
                                 (i & (cw - 1))) = *reinterpret_cast<const U16x4 *>(s.pred + i);
  }
  const int nnz =
      sq16 ? tx2_job<TXM, 64, RDOQ, 16, 16>(s.tx, tb, 3 * ci, bd, orig.c[0], s.pred, cw, rec.c[0],
                                            FWD ? coeffs : nullptr, FWD ? coeff_off : nullptr,
                                            nnz_out, tx_tables, tx_tables_t, lay, &orig_pre, 0,
                                            rq_wave, rq_ctx, rq_prm, nullptr, FWD ? &fcl : nullptr)
           : tx2_job<TXM, 64, RDOQ>(s.tx, tb, 3 * ci, bd, orig.c[0], s.pred, cw, rec.c[0],
                                    FWD ? coeffs : nullptr, FWD ? coeff_off : nullptr, nnz_out,
                                    tx_tables, tx_tables_t, lay, &orig_pre, 0, rq_wave, rq_ctx,
                                    rq_prm, nullptr, FWD ? &fcl : nullptr);
  ME2_TRACE(8);
  // (FWD: the record with cbf_luma = 0; the quantiser's walk sets the flag of the
  // blocks it codes a level for, quant_rdo_packed_wave's cu_patch)
  if (cus && ME2_LANE == 0) {
    xvcgpu_cu_info c;
    c.x = (uint16_t)mb.x;
    c.y = (uint16_t)mb.y;
    c.w = mb.w;
    c.h = mb.h;
    c.intra = 0;
    c.cbf_luma = !FWD && nnz != 0;
    c.qp_y = (int8_t)qp_y;
    c.qp_c = (int8_t)qp_c;
    c.ref_idx0 = 0;
    c.reserved = 0;
    c.ref_poc[0] = ref_poc;
    c.ref_poc[1] = -1;
    for (int k = 0; k < 4; k++) {
      c.mv[0][k][0] = mr.mv_x;
      c.mv[0][k][1] = mr.mv_y;
      c.mv[1][k][0] = 0;
      c.mv[1][k][1] = 0;
    }
    cus[ci] = c;
  }
}

// Waves per SIMD the forward-only instance is compiled for (LDS allows 7: 22 KB per workgroup
// of four waves).  Registers / spilled dwords: 5: 93 / 0, 6: 80 / 68, 7: 72 / 104 - with the
// exact 16x16 / 8x8 instances marked likely (above) the spills all sit in the any-size path.
// Measured 1080p / 2160p / 4320p passes/s: 6: 8605 / 1928 / 745, 7: 8656 / 1943 / 768 (before the
// likely marks the spills sat at the kernel's entry: 5: 8220 / 1818 / 688, 6: 8288 / 1837 /
// 698, 7: 7712 / 1667 / 607).
#ifndef RECON_FWD_MIN_WAVES
#define RECON_FWD_MIN_WAVES 7
#endif
template <bool RDOQ = false, bool FWD = false>
__global__ void __launch_bounds__(256, FWD && !RDOQ ? RECON_FWD_MIN_WAVES : 1)
recon_from_me_kernel(PicView orig, PicView ref, PicView rec, const xvcgpu_me_block *blocks, const xvcgpu_me_result *results, int n_cus, int qp_y, int qp_c, int intra_pic, int ref_poc, int32_t *nnz_out, xvcgpu_cu_info *cus, const int16_t *tx_tables, const int16_t *tx_tables_t, TxTableLayout lay, const xvcgpu_rdoq_contexts *rq_ctx = nullptr, const xvcgpu_rdoq_params *rq_prm = nullptr, int16_t *coeffs = nullptr, const uint32_t *coeff_off = nullptr, FwdClassify fc = FwdClassify()) {
  recon_from_me_kernel_body<RDOQ, FWD>(orig, ref, rec, blocks, results, n_cus, qp_y, qp_c, intra_pic, ref_poc, nnz_out, cus, tx_tables, tx_tables_t, lay, rq_ctx, rq_prm, coeffs, coeff_off, fc);
}

#endif  // XVCGPU_K_RECON_H_
