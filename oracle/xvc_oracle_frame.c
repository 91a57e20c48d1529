/*
 * xvc_oracle_frame.c -- CPU restatement of one "hot-path frame pass": the
 * composition bench.py times on the GPU (DESIGN.md section "Frame pass"):
 *   for every CU: TZ full-pel search + sub-pel refinement (T1,T3),
 *                 motion compensation Y,U,V (I1),
 *                 residual -> transform -> quant -> dequant -> inverse -> rec
 *                 (X1,Q,Q1,X2,R1)
 *   then: deblocking (D1-D4), border extension (P1), picture SSD (M6).
 *
 * TEST INFRASTRUCTURE ONLY (see xvc_oracle.h): used as the checker of the GPU
 * frame pass and as bench.py's `cpu_baseline` fallback (kind "port"; where the
 * reference build is present bench.py times xr_frame_pass instead).  Each step calls
 * the pinned block-level oracle functions; nothing new is computed here.
 */
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"


void xo_frame_pass(xo_frame_args *a) {
  const int bd = a->bd;
  const int nthreads = a->threads > 1 ? a->threads : 1;
  (void)nthreads;
  (void)xo_transform_matrix(XVC_TX_DCT2, 4); /* lazily built tables: before
                                                 any parallel region */
  /* motion search + motion compensation: CUs are independent */
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
  for (int i = 0; i < a->n_cus; i++) {
    const xvcgpu_me_block *b = &a->me_blocks[i];
    xvcgpu_me_result *r = &a->me_results[i];
    int fp[2], mv[2];
    uint32_t cost = 0, dist = 0;
    xo_tz_search(bd, b, a->pic_w, a->pic_h, a->orig[0], a->orig_stride[0],
                 a->ref[0], a->ref_stride[0], fp, &cost);
    if (b->fullpel_mv) {
      mv[0] = fp[0] * 16;
      mv[1] = fp[1] * 16;
    } else {
      xo_subpel_search(bd, b, a->pic_w, a->pic_h, a->orig[0], a->orig_stride[0],
                       a->ref[0], a->ref_stride[0], fp, mv, &dist);
    }
    r->fullpel_x = fp[0];
    r->fullpel_y = fp[1];
    r->fullpel_cost = cost;
    r->mv_x = mv[0];
    r->mv_y = mv[1];
    r->subpel_dist = dist;
    for (int c = 0; c < 3; c++) {
      const int cs = c ? 1 : 0;
      uint16_t *dst = a->pred[c] + (ptrdiff_t)(b->y >> cs) * a->pred_stride[c] +
                      (b->x >> cs);
      xo_mc_block(bd, c, b->x, b->y, b->w, b->h, mv[0], mv[1], a->pic_w,
                  a->pic_h, a->ref[c], a->ref_stride[c], dst, a->pred_stride[c]);
    }
  }
  /* residual pipeline */
#pragma omp parallel num_threads(nthreads)
  {
    int16_t *levels = (int16_t *)malloc(sizeof(int16_t) * 64 * 64);
#pragma omp for schedule(dynamic, 48)
    for (int i = 0; i < a->n_tx; i++) {
      const xvcgpu_tx_block *t = &a->tx_blocks[i];
      const int c = t->comp;
      if (a->rdoq_params)
        a->nnz[i] = xo_residual_pipeline_rdoq(bd, t, a->rdoq_contexts, &a->rdoq_params[i],
                                              a->orig[c], a->orig_stride[c], a->pred[c],
                                              a->pred_stride[c], a->rec[c], a->rec_stride[c],
                                              levels);
      else
        a->nnz[i] = xo_residual_pipeline(bd, t, a->orig[c], a->orig_stride[c],
                                         a->pred[c], a->pred_stride[c], a->rec[c],
                                         a->rec_stride[c], levels);
    }
    free(levels);
  }
  /* CU metadata for the in-loop filter */
  for (int i = 0; i < a->n_cus; i++) {
    const xvcgpu_me_block *b = &a->me_blocks[i];
    xvcgpu_cu_info *c = &a->cus[a->cu_base + i];
    memset(c, 0, sizeof(*c));
    c->x = (uint16_t)b->x;
    c->y = (uint16_t)b->y;
    c->w = b->w;
    c->h = b->h;
    c->cbf_luma = a->nnz[a->luma_tx_index ? a->luma_tx_index[i] : i] != 0;
    c->qp_y = (int8_t)a->qp_y;
    c->qp_c = (int8_t)a->qp_c;
    c->ref_poc[0] = a->ref_poc;
    c->ref_poc[1] = -1;
    for (int k = 0; k < 4; k++) {
      c->mv[0][k][0] = a->me_results[i].mv_x;
      c->mv[0][k][1] = a->me_results[i].mv_y;
    }
  }
  if (a->encode_only) return;
  xo_deblock_picture(bd, a->pic_w, a->pic_h, 0, a->beta_offset, a->tc_offset,
                     a->subblock, a->cus, a->cu_map, a->map_stride, a->rec,
                     a->rec_stride);
  for (int c = 0; c < 3; c++) {
    const int cs = c ? 1 : 0;
    xo_pad_border(a->pic_w >> cs, a->pic_h >> cs, a->border[c], a->border[c],
                  a->rec[c], a->rec_stride[c]);
  }
  a->ssd[0] = xo_picture_ssd(bd, a->pic_w, a->pic_h, a->orig[0], a->orig_stride[0],
                             a->rec[0], a->rec_stride[0], NULL, &a->ssd[1]);
}
