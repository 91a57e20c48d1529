// k_inter_pred.h -- the decoder's inter prediction for any inter CU (SURVEY 8f
// N1): InterPrediction::MotionCompensation (inter_prediction.cc:710-738) ->
// MotionCompRefList (:1012-1042) -> MotionCompUniPred -> Sample / int16
// (:1138-1172), MotionCompAffine for either output type (:1044-1136),
// LocalIlluminationComp (:1555-1575, DeriveLicParams :1577-1663), AddAvgBi
// (:1540-1553).  One workgroup of 256 per (CU, component) job; the reference
// pictures of the batch come as a table indexed by the job's per-list slot, so
// one launch covers all reference pictures of both lists.
//
// Translational lists use the workgroup filters (k_interp.h, k_bipred.h:
// separable, intermediate in LDS).  Affine lists give every thread one output
// sample with its sub-block's own vector and filter phase and evaluate the two
// filter stages directly (d_interp_point: identical rounding, stage by stage);
// a one-4x4-sub-block-at-a-time walk would leave 15/16 of the workgroup idle.
#ifndef XVCGPU_K_INTER_PRED_H_
#define XVCGPU_K_INTER_PRED_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_bipred.h"
#include "k_interp.h"
#include "xvcgpu_internal.h"

// One output sample of MotionCompUniPred at full-pel pointer `s` (the sample's
// own position displaced by the integer vector): BIPRED = false -> Sample
// (FilterLuma / FilterChroma, :1387-1448), true -> the 14-bit int16
// intermediate (Filter*Bipred / FilterCopyBipred, :1450-1538).
template <bool CHROMA, bool BIPRED>
__device__ __forceinline__ int d_interp_point(int bd, int fx, int fy, const uint16_t *s, int rs) {
  constexpr int N = CHROMA ? 4 : 8;
  constexpr int BACK = N / 2 - 1;
  const int16_t *fh = CHROMA ? kChromaTaps[fx] : kLumaTaps[fx];
  const int16_t *fv = CHROMA ? kChromaTaps[fy] : kLumaTaps[fy];
  const int smax = (1 << bd) - 1, head = 14 - bd;
  const int sh1 = 6 - head, off1 = -(8192 << sh1);
  if (fx == 0 && fy == 0) {
    if (BIPRED) return (int16_t)((int16_t)(s[0] << head) - (int16_t)8192);
    return s[0];
  }
  if (fy == 0 || fx == 0) {
    const int16_t *f = fy == 0 ? fh : fv;
    const ptrdiff_t step = fy == 0 ? 1 : rs;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < N; k++) sum += (int)s[(k - BACK) * step] * f[k];
    if (BIPRED) return (int16_t)((sum + off1) >> sh1);
    // the vertical Sample filter narrows to int16 before the clip (:1290)
    return fy == 0 ? d_clip_bd((sum + 32) >> 6, smax) : d_clip_bd((int16_t)((sum + 32) >> 6), smax);
  }
  int acc = 0;
#pragma unroll
  for (int r = 0; r < N; r++) {
    const uint16_t *row = s + (ptrdiff_t)(r - BACK) * rs - BACK;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < N; k++) sum += (int)row[k] * fh[k];
    acc += (int)(int16_t)((sum + off1) >> sh1) * fv[r];
  }
  if (BIPRED) return (int16_t)(acc >> 6);
  const int sh2 = 6 + head, off2 = (8192 << 6) + (1 << (sh2 - 1));
  return d_clip_bd((int16_t)((acc + off2) >> sh2), smax);
}

// MotionCompAffine of the whole block by the workgroup: out[y * os + x] for the
// cw x ch component block (Sample: uint16 view, BIPRED: int16 values stored in
// the same 16-bit cells).
template <bool BIPRED>
__device__ __forceinline__ void wg_affine_block(int bd, int comp, int bx, int by, int bw, int bh,
                                                const int32_t (*mv_in)[2], int pic_w, int pic_h,
                                                const PlaneView &pr, uint16_t *out, int os) {
  int mv[3][2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    mv[i][0] = mv_in[i][0];
    mv[i][1] = mv_in[i][1];
    d_clip_mv(bx, by, pic_w, pic_h, mv[i][0], mv[i][1]);
  }
  const int cs = comp ? 1 : 0, shift = 4 + cs, mask = (1 << shift) - 1;
  const int cx = bx >> cs, cy = by >> cs, cw = bw >> cs, ch = bh >> cs;
  const int lw = 31 - __clz(cw);
  const bool plain = mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1];
  int sbw = cw, sbh = ch;
  if (!plain) {
    sbw = d_affine_subblock(mv[0][0], mv[0][1], mv[1][0], mv[1][1], cw, cs);
    sbh = d_affine_subblock(mv[0][0], mv[0][1], mv[2][0], mv[2][1], ch, cs);
  }
  const int mv_max_x = (pic_w - bx + 8 - 1) * 16, mv_min_x = (-64 - bx - 8 + 1) * 16;
  const int mv_max_y = (pic_h - by + 8 - 1) * 16, mv_min_y = (-64 - by - 8 + 1) * 16;
  const int dhx = ((mv[1][0] - mv[0][0]) * 256) / cw;  // C division
  const int dhy = ((mv[1][1] - mv[0][1]) * 256) / cw;
  const int dvx = -dhy, dvy = dhx;
  for (int i = threadIdx.x; i < cw * ch; i += blockDim.x) {
    const int y = i >> lw, x = i & (cw - 1);
    int mx = mv[0][0], my = mv[0][1];
    if (!plain) {
      const int ix = x / sbw, iy = y / sbh;
      // the reference's running sums in closed form (:1103-1133)
      const int hor_x = mv[0][0] * 256 + dvx * sbh * iy + dhx * sbw * ix;
      const int hor_y = mv[0][1] * 256 + dvy * sbh * iy + dhy * sbw * ix;
      mx = d_clip3((hor_x + dhx * (sbw >> 1) + dvx * (sbh >> 1)) >> 8, mv_min_x, mv_max_x);
      my = d_clip3((hor_y + dhy * (sbw >> 1) + dvy * (sbh >> 1)) >> 8, mv_min_y, mv_max_y);
    }
    const uint16_t *s = pr.p + (ptrdiff_t)(cy + y + (my >> shift)) * pr.stride + cx + x + (mx >> shift);
    const int v = comp ? d_interp_point<true, BIPRED>(bd, mx & mask, my & mask, s, pr.stride)
                       : d_interp_point<false, BIPRED>(bd, mx & mask, my & mask, s, pr.stride);
    out[y * os + x] = (uint16_t)v;
  }
}

// (wg_lic_model: k_bipred.h)

// grid: n; block: 256.
__device__ __forceinline__ void
inter_pred_body(const RefTable &refs, const PicView &rec, const PicView &pred,
                const xvcgpu_inter_block *blocks, int n, const xvcgpu_block_pos *dst_pos,
                int pic_w, int pic_h) {
  __shared__ int16_t tmp[64 * 71];
  __shared__ int16_t p16[2][64 * 64];
  __shared__ uint16_t smp[64 * 64];
  __shared__ int s_scale, s_offset;
  const int bi_ = blockIdx.x;
  if (bi_ >= n) return;
  const xvcgpu_inter_block &b = blocks[bi_];
  const int bd = pred.bd, comp = b.comp;
  const int cs = comp ? 1 : 0, shift = 4 + cs, mask = (1 << shift) - 1;
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  const bool affine = b.flags & XVC_INTER_AFFINE;
  const bool lic = (b.flags & XVC_INTER_LIC) && !affine;
  const bool bi = b.ref[0] >= 0 && b.ref[1] >= 0;
  const PlaneView pd = pred.c[comp];
  // the CU's position, or the caller's scratch position (pred is then not a picture
  // of the sequence)
  const int ox = dst_pos ? dst_pos[bi_].x >> cs : cx, oy = dst_pos ? dst_pos[bi_].y >> cs : cy;
  uint16_t *out = pd.p + (ptrdiff_t)oy * pd.stride + ox;
  const int smax = (1 << bd) - 1, head = 14 - bd;
  const int lw = 31 - __clz(cw);
  for (int l = 0; l < 2; l++) {
    if (b.ref[l] < 0) continue;  // uniform over the workgroup
    const PlaneView pr = refs.pic[b.ref[l]].c[comp];
    // the list's vectors straight from the descriptor in global memory (indexing a
    // register copy by the run-time list puts it in scratch)
    const int32_t (*mvl)[2] = blocks[bi_].mv[l];
    if (bi && !lic) {  // normal bi-prediction: the list at 14 bit
      if (affine) {
        wg_affine_block<true>(bd, comp, b.x, b.y, b.w, b.h, mvl, pic_w, pic_h, pr,
                              reinterpret_cast<uint16_t *>(p16[l]), cw);
      } else {
        int mx = mvl[0][0], my = mvl[0][1];
        d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);
        const uint16_t *r = pr.p + (ptrdiff_t)(cy + (my >> shift)) * pr.stride + cx + (mx >> shift);
        __syncthreads();  // tmp reuse
        if (comp)
          wg_interp_block_bipred<true>(bd, cw, ch, mx & mask, my & mask, r, pr.stride, tmp, p16[l]);
        else
          wg_interp_block_bipred<false>(bd, cw, ch, mx & mask, my & mask, r, pr.stride, tmp, p16[l]);
      }
      continue;
    }
    // Sample prediction of this list: straight to the output (uni-pred) or to
    // LDS (bi-pred with LIC, :725-731)
    uint16_t *dst = bi ? smp : out;
    const int ds = bi ? cw : pd.stride;
    if (affine) {
      wg_affine_block<false>(bd, comp, b.x, b.y, b.w, b.h, mvl, pic_w, pic_h, pr, dst, ds);
      continue;
    }
    int mx = mvl[0][0], my = mvl[0][1];
    d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);
    const uint16_t *r = pr.p + (ptrdiff_t)(cy + (my >> shift)) * pr.stride + cx + (mx >> shift);
    __syncthreads();  // tmp / smp reuse
    if (comp)
      wg_interp_block<true>(bd, cw, ch, mx & mask, my & mask, r, pr.stride, tmp, dst, ds);
    else
      wg_interp_block<false>(bd, cw, ch, mx & mask, my & mask, r, pr.stride, tmp, dst, ds);
    if (!lic) continue;
    wg_lic_model(bd, comp, b.x, b.y, b.w, b.h, mx, my, b.neighbors, b.above_x, b.above_y,
                 b.left_x, b.left_y, pic_w, pic_h, pr, rec.c[comp], &s_scale, &s_offset);
    __syncthreads();  // model visible; prediction stores ordered before the reads below
    const int scale = s_scale, offset = s_offset;
    for (int i = threadIdx.x; i < cw * ch; i += 256) {
      uint16_t *p = dst + (i >> lw) * ds + (i & (cw - 1));
      const int v = d_clip3(((scale * (int)*p) >> 5) + offset, 0, smax);
      if (bi)  // FilterCopyBipred of the compensated sample (:728, :730)
        p16[l][i] = (int16_t)((int16_t)(v << head) - (int16_t)8192);
      else
        *p = (uint16_t)v;
    }
  }
  if (!bi) return;
  __syncthreads();
  // AddAvgBi (inter_prediction.cc:1545-1547)
  const int sh = (head > 2 ? head : 2) + 1;
  const int off = (1 << (sh - 1)) + 2 * 8192;
  for (int i = threadIdx.x; i < cw * ch; i += 256)
    out[(ptrdiff_t)(i >> lw) * pd.stride + (i & (cw - 1))] =
        d_clip_bd(((int)p16[0][i] + (int)p16[1][i] + off) >> sh, smax);
}

__global__ void __launch_bounds__(256)
inter_pred_kernel(RefTable refs, PicView rec, PicView pred, const xvcgpu_inter_block *blocks,
                  int n, const xvcgpu_block_pos *dst_pos, int pic_w, int pic_h) {
  inter_pred_body(refs, rec, pred, blocks, n, dst_pos, pic_w, pic_h);
}

#endif  // XVCGPU_K_INTER_PRED_H_
