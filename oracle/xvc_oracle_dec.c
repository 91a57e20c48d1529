/*
 * xvc_oracle_dec.c -- CPU restatement of the decoder's reconstruction stage.
 *
 * TEST INFRASTRUCTURE (the oracle): never linked or imported by the product.
 * Pinned against the reference decoder by tests/test_stream_oracle.py: the
 * parsed syntax of real streams (tests/golden/stream_*.npz, captured from the
 * reference decoder) must reconstruct to the planes / MD5 the reference
 * produced.
 *
 * Restates, CU by CU in coding order:
 *   CuDecoder::DecompressCu / DecompressComponent  xvc_dec_lib/cu_decoder.cc:84-138
 *   CuDecoder::PredictIntra                        :140-147
 *   IntraPrediction::DetermineNeighbors            xvc_common_lib/intra_prediction.cc:688-705
 *   CodingUnit::GetCuSizeAboveRight / BelowLeft    xvc_common_lib/coding_unit.cc:304-336
 *   InterPrediction::MotionCompensation            xvc_common_lib/inter_prediction.cc:710-738
 *   InterPrediction::MotionCompRefList             :1012-1042
 *   InterPrediction::MotionCompAffine -> int16     :1044-1136
 *   PictureDecoder::Decode tail (deblock, pad)     xvc_dec_lib/picture_decoder.cc:186-196
 *   DeblockingFilter::DeblockPicture, two CU trees xvc_common_lib/deblocking_filter.cc:56-77
 */
#include <stdlib.h>
#include <string.h>

#include "../include/xvc_syntax.h"
#include "xvc_oracle.h"

static int xd_clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* get_subblock_size lambda, inter_prediction.cc:1071-1086 */
static int xd_affine_subblock(int ref_x, int ref_y, int mv_x, int mv_y, int size, int scale) {
  const int dx = abs(mv_x - ref_x), dy = abs(mv_y - ref_y);
  const int max_len = dx > dy ? dx : dy;
  if (!max_len) return size;
  int sb = (size >> 2) / max_len;
  if (sb < 1) sb = 1;
  while (size % sb) sb--;
  return (sb > 4 ? sb : 4) >> scale;
}

/* MotionCompAffine<DataBuffer<int16_t>> (inter_prediction.cc:1044-1136): as
 * xo_mc_affine_block, every sub-block through the int16 MotionCompUniPred */
static void xd_mc_affine_i16(int bd, int comp, int x, int y, int w, int h, const int32_t mv_in[3][2],
                             int pic_w, int pic_h, const uint16_t *ref_plane, ptrdiff_t rs,
                             int16_t *pred, ptrdiff_t ps) {
  int mv[3][2];
  for (int i = 0; i < 3; i++) {
    mv[i][0] = mv_in[i][0];
    mv[i][1] = mv_in[i][1];
    xo_clip_mv(x, y, pic_w, pic_h, &mv[i][0], &mv[i][1]);
  }
  const int cs = comp ? 1 : 0, shift = 4 + cs, mask = (1 << shift) - 1;
  const int cx = x >> cs, cy = y >> cs, cw = w >> cs, ch = h >> cs;
  if (mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) {
    const uint16_t *r =
        ref_plane + (ptrdiff_t)(cy + (mv[0][1] >> shift)) * rs + cx + (mv[0][0] >> shift);
    xo_mc_uni_bipred(bd, comp != 0, cw, ch, mv[0][0] & mask, mv[0][1] & mask, r, rs, pred, ps);
    return;
  }
  const int sbw = xd_affine_subblock(mv[0][0], mv[0][1], mv[1][0], mv[1][1], cw, cs);
  const int sbh = xd_affine_subblock(mv[0][0], mv[0][1], mv[2][0], mv[2][1], ch, cs);
  const int mv_max_x = (pic_w - x + 8 - 1) * 16, mv_min_x = (-64 - x - 8 + 1) * 16;
  const int mv_max_y = (pic_h - y + 8 - 1) * 16, mv_min_y = (-64 - y - 8 + 1) * 16;
  const int dhx = ((mv[1][0] - mv[0][0]) * 256) / cw;
  const int dhy = ((mv[1][1] - mv[0][1]) * 256) / cw;
  const int dvx = -dhy, dvy = dhx;
  int hor_x = mv[0][0] * 256, hor_y = mv[0][1] * 256;
  int ver_x = hor_x, ver_y = hor_y;
  for (int sy = 0; sy < ch; sy += sbh) {
    for (int sx = 0; sx < cw; sx += sbw) {
      int mx = (hor_x + dhx * (sbw >> 1) + dvx * (sbh >> 1)) >> 8;
      int my = (hor_y + dhy * (sbw >> 1) + dvy * (sbh >> 1)) >> 8;
      mx = xd_clip3(mx, mv_min_x, mv_max_x);
      my = xd_clip3(my, mv_min_y, mv_max_y);
      const uint16_t *r =
          ref_plane + (ptrdiff_t)(cy + sy + (my >> shift)) * rs + cx + sx + (mx >> shift);
      xo_mc_uni_bipred(bd, comp != 0, sbw, sbh, mx & mask, my & mask, r, rs,
                       pred + (ptrdiff_t)sy * ps + sx, ps);
      hor_x += dhx * sbw;
      hor_y += dhy * sbw;
    }
    ver_x += dvx * sbh;
    ver_y += dvy * sbh;
    hor_x = ver_x;
    hor_y = ver_y;
  }
}

/* InterPrediction::MotionCompensation for one component of one CU
 * (inter_prediction.cc:710-738).  ref_planes[slot * 3 + comp] / rec_planes[comp]
 * / pred_planes[comp] point at sample (0,0); all pictures share `strides`. */
void xo_inter_pred_block(int bd, const xvcgpu_inter_block *b, int pic_w, int pic_h,
                         const uint16_t *const *ref_planes, const uint16_t *const *rec_planes,
                         uint16_t *const *pred_planes, const ptrdiff_t *strides) {
  const int c = b->comp, cs = c ? 1 : 0;
  const ptrdiff_t st = strides[c];
  const int cx = b->x >> cs, cy = b->y >> cs, cw = b->w >> cs, ch = b->h >> cs;
  const int affine = b->flags & XVC_INTER_AFFINE, lic = (b->flags & XVC_INTER_LIC) && !affine;
  const int bi = b->ref[0] >= 0 && b->ref[1] >= 0;
  uint16_t *out = pred_planes[c] + (ptrdiff_t)cy * st + cx;
  int16_t p16[2][64 * 64];
  uint16_t tmp[64 * 64];
  for (int l = 0; l < 2; l++) {
    if (b->ref[l] < 0) continue;
    const uint16_t *ref = ref_planes[b->ref[l] * 3 + c];
    if (bi && !lic) { /* normal bi-prediction: both lists at 14 bit */
      if (affine) {
        xd_mc_affine_i16(bd, c, b->x, b->y, b->w, b->h, b->mv[l], pic_w, pic_h, ref, st, p16[l],
                         64);
      } else {
        int mx = b->mv[l][0][0], my = b->mv[l][0][1];
        xo_clip_mv(b->x, b->y, pic_w, pic_h, &mx, &my);
        const int shift = 4 + cs, mask = (1 << shift) - 1;
        const uint16_t *r = ref + (ptrdiff_t)(cy + (my >> shift)) * st + cx + (mx >> shift);
        xo_mc_uni_bipred(bd, c != 0, cw, ch, mx & mask, my & mask, r, st, p16[l], 64);
      }
      continue;
    }
    /* Sample prediction of this list: into the output (uni) or a temporary */
    uint16_t *dst = bi ? tmp : out;
    const ptrdiff_t ds = bi ? 64 : st;
    if (affine) {
      int mv[3][2];
      for (int i = 0; i < 3; i++) {
        mv[i][0] = b->mv[l][i][0];
        mv[i][1] = b->mv[l][i][1];
      }
      xo_mc_affine_block(bd, c, b->x, b->y, b->w, b->h, (const int(*)[2])mv, pic_w, pic_h, ref,
                         st, dst, ds);
    } else if (lic) {
      /* xo_mc_lic_block addresses its output by the block position */
      xvcgpu_mc_lic_block lb;
      memset(&lb, 0, sizeof(lb));
      lb.x = b->x;
      lb.y = b->y;
      lb.w = b->w;
      lb.h = b->h;
      lb.comp = b->comp;
      lb.neighbors = b->neighbors;
      lb.mv_x = b->mv[l][0][0];
      lb.mv_y = b->mv[l][0][1];
      lb.above_x = b->above_x;
      lb.above_y = b->above_y;
      lb.left_x = b->left_x;
      lb.left_y = b->left_y;
      xo_mc_lic_block(bd, &lb, pic_w, pic_h, ref, st, rec_planes[c], st,
                      dst - ((ptrdiff_t)cy * ds + cx), ds);
    } else {
      xo_mc_block(bd, c, b->x, b->y, b->w, b->h, b->mv[l][0][0], b->mv[l][0][1], pic_w, pic_h,
                  ref, st, dst, ds);
    }
    if (bi) { /* FilterCopyBipred of the compensated samples (:728, :730) */
      xo_mc_uni_bipred(bd, c != 0, cw, ch, 0, 0, tmp, 64, p16[l], 64);
    }
  }
  if (bi) xo_add_avg(bd, cw, ch, p16[0], 64, p16[1], 64, out, st);
}

/* ---- neighbour availability --------------------------------------------- */
typedef struct {
  int32_t *cell[2]; /* per tree: coding-order index of the CU covering a 4x4 cell, -1 */
  int stride, rows;
} xd_map;

static int xd_cell(const xd_map *m, int tree, int x, int y) {
  if (x < 0 || y < 0) return -1;
  const int cx = x >> 2, cy = y >> 2;
  if (cx >= m->stride || cy >= m->rows) return -1;
  return m->cell[tree][cy * m->stride + cx];
}

/* IntraPrediction::DetermineNeighbors with CodingUnit::GetCuSizeAboveRight /
 * GetCuSizeBelowLeft, evaluated on the cells marked so far */
void xo_intra_neighbors(const int32_t *cells, int stride, int rows, const xvc_cu_syntax *cu,
                        int comp, uint8_t *flags, uint8_t *above_right, uint8_t *below_left) {
  xd_map m;
  m.cell[0] = m.cell[1] = (int32_t *)cells;
  m.stride = stride;
  m.rows = rows;
  const int cs = comp ? 1 : 0;
  const int x = cu->x >> cs, y = cu->y >> cs;
  int f = 0, ar = 0, bl = 0;
  if (x > 0) {
    f |= XVC_INTRA_HAS_LEFT;
    const int px = cu->x - 4, py = cu->y + cu->h - 4;
    for (int i = cu->w; i >= 0; i -= 4)
      if (xd_cell(&m, 0, px, py + i) >= 0) {
        bl = i >> cs;
        break;
      }
  }
  if (y > 0) {
    f |= XVC_INTRA_HAS_ABOVE;
    const int px = cu->x + cu->w - 4, py = cu->y - 4;
    for (int i = cu->h; i >= 0; i -= 4)
      if (xd_cell(&m, 0, px + i, py) >= 0) {
        ar = i >> cs;
        break;
      }
  }
  if (x > 0 && y > 0) f |= XVC_INTRA_HAS_ABOVE_LEFT;
  *flags = (uint8_t)f;
  *above_right = (uint8_t)ar;
  *below_left = (uint8_t)bl;
}

/* ---- one picture ---------------------------------------------------------- */
/* planes[c] / ref_planes[slot * 3 + c] point at sample (0,0) of padded
 * pictures that all share `strides` and have at least `border` (luma; half for
 * chroma) samples around them.  ref_slot[list][idx] = slot of that reference.
 * nb_out (optional, 9 bytes per CU): the neighbour state each intra CU saw
 * (flags, above_right, below_left per component) - compared by the tests with
 * what the reference's IntraPrediction::DetermineNeighbors reported.
 * pre_planes (optional): copy of the visible area before the in-loop filter
 * (tightly packed planes). */
void xo_decode_picture(const xvc_picture_syntax *ps, const xvc_cu_syntax *cus,
                       const int16_t *levels, const uint16_t *const *ref_planes,
                       const int32_t ref_slot[2][5], uint16_t *const planes[3],
                       const ptrdiff_t strides[3], int border, uint8_t *nb_out,
                       uint16_t *const pre_planes[3]) {
  const int pic_w = ps->width, pic_h = ps->height, bd = ps->bitdepth;
  const int smax = (1 << bd) - 1;
  xd_map m;
  m.stride = (pic_w + 63) / 4 + 1;
  m.rows = (pic_h + 63) / 4 + 1;
  const size_t cells = (size_t)m.stride * m.rows;
  m.cell[0] = (int32_t *)malloc(cells * sizeof(int32_t));
  m.cell[1] = (int32_t *)malloc(cells * sizeof(int32_t));
  memset(m.cell[0], 0xff, cells * sizeof(int32_t));
  memset(m.cell[1], 0xff, cells * sizeof(int32_t));
  uint16_t *pred = (uint16_t *)malloc(sizeof(uint16_t) * 64 * 64);
  int two_trees = 0;

  for (int i = 0; i < ps->n_cus; i++) {
    const xvc_cu_syntax *cu = &cus[i];
    const int tree = cu->tree;
    if (tree) two_trees = 1;
    /* PictureData::MarkUsedInPic (picture_data.cc:191-210) */
    for (int yy = 0; yy < cu->h; yy += 4)
      for (int xx = 0; xx < cu->w; xx += 4)
        m.cell[tree][((cu->y + yy) >> 2) * m.stride + ((cu->x + xx) >> 2)] = i;
    /* GetComponents(cu_tree): an intra picture codes luma and chroma in two
     * trees (picture_data.cc:71-76); otherwise one tree carries Y, U, V */
    const int c0 = tree ? 1 : 0;
    const int c1 = tree ? 3 : (ps->pic_type == XVC_PIC_INTRA ? 1 : 3);
    for (int c = c0; c < c1; c++) {
      const int cs = c ? 1 : 0;
      const int x = cu->x >> cs, y = cu->y >> cs, w = cu->w >> cs, h = cu->h >> cs;
      uint16_t *dst = planes[c] + (ptrdiff_t)y * strides[c] + x;
      /* prediction into `pred` (stride 64) */
      if (cu->pred_mode == 0) {
        xvcgpu_intra_block jb;
        memset(&jb, 0, sizeof(jb));
        jb.x = (int16_t)x;
        jb.y = (int16_t)y;
        jb.w = (uint8_t)w;
        jb.h = (uint8_t)h;
        jb.comp = (uint8_t)c;
        xd_map one = m;
        one.cell[0] = m.cell[tree];
        xo_intra_neighbors(one.cell[0], m.stride, m.rows, cu, c, &jb.neighbors, &jb.above_right,
                           &jb.below_left);
        if (nb_out) {
          nb_out[9 * i + c] = jb.neighbors;
          nb_out[9 * i + 3 + c] = jb.above_right;
          nb_out[9 * i + 6 + c] = jb.below_left;
        }
        if (cu->intra_mode[c] == XVC_CU_INTRA_LM) {
          xo_intra_lm_chroma(bd, x, y, w, h, planes[0], strides[0], planes[c], strides[c], pred,
                             64);
        } else {
          jb.mode = (uint8_t)cu->intra_mode[c];
          /* xo_intra_pred_block writes at the block's position of its output */
          xo_intra_pred_block(bd, &jb, planes[c], strides[c], pred - ((ptrdiff_t)y * 64 + x), 64);
        }
      } else {
        xvcgpu_inter_block ib;
        memset(&ib, 0, sizeof(ib));
        ib.x = cu->x;
        ib.y = cu->y;
        ib.w = cu->w;
        ib.h = cu->h;
        ib.comp = (uint8_t)c;
        ib.flags = (uint8_t)(((cu->flags & XVC_CU_AFFINE) ? XVC_INTER_AFFINE : 0) |
                             ((cu->flags & XVC_CU_LIC) ? XVC_INTER_LIC : 0));
        for (int l = 0; l < 2; l++) {
          const int used = cu->inter_dir == 2 || cu->inter_dir == l;
          ib.ref[l] = (int8_t)(used ? ref_slot[l][cu->ref_idx[l]] : -1);
          memcpy(ib.mv[l], cu->mv[l], sizeof(ib.mv[l]));
        }
        /* LIC: GetCodingUnitAbove / Left (coding_unit.cc:227-234, :275-282) */
        const int ia = cu->y > 0 ? xd_cell(&m, tree, cu->x, cu->y - 4) : -1;
        const int il = cu->x > 0 ? xd_cell(&m, tree, cu->x - 4, cu->y) : -1;
        if (ia >= 0) {
          ib.neighbors |= XVC_LIC_HAS_ABOVE;
          ib.above_x = cus[ia].x;
          ib.above_y = cus[ia].y;
        }
        if (il >= 0) {
          ib.neighbors |= XVC_LIC_HAS_LEFT;
          ib.left_x = cus[il].x;
          ib.left_y = cus[il].y;
        }
        /* xo_inter_pred_block uses one stride set for references, reconstruction
         * and prediction: predict straight into the picture (the LIC model reads
         * only the row above / column left of the block), then move */
        xo_inter_pred_block(bd, &ib, pic_w, pic_h, ref_planes, (const uint16_t *const *)planes,
                            planes, strides);
        for (int yy = 0; yy < h; yy++) memcpy(pred + yy * 64, dst + (ptrdiff_t)yy * strides[c], 2 * w);
      }
      if (!cu->cbf[c]) { /* cu_decoder.cc:110, :120-122 */
        for (int yy = 0; yy < h; yy++) memcpy(dst + (ptrdiff_t)yy * strides[c], pred + yy * 64, 2 * w);
        continue;
      }
      /* Quantize::Inverse -> InverseTransform -> AddClip (cu_decoder.cc:124-137) */
      int16_t deq[64 * 64], resi[64 * 64];
      const int16_t *lv = levels + cu->level_off[c];
      xo_dequant(bd, cu->qp[c], w, h, lv, w, deq, 64);
      if (cu->tx_skip[c]) {
        xo_inv_transform_skip(bd, w, h, deq, 64, resi, 64);
      } else {
        int nnz = 0;
        for (int k = 0; k < w * h; k++) nnz += lv[k] != 0;
        const int dc_only = nnz == 1 && lv[0] != 0;
        /* can_dst_4x4 (transform.cc:88-90) */
        const int dst4 = c == 0 && cu->pred_mode == 0 && cu->tx_type[c][0] == XVC_TX_DEFAULT &&
                         cu->tx_type[c][1] == XVC_TX_DEFAULT;
        xo_inv_transform(bd, w, h, cu->tx_type[c][1], cu->tx_type[c][0], dst4, dc_only, deq, 64,
                         resi, 64);
      }
      for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++)
          dst[(ptrdiff_t)yy * strides[c] + xx] =
              (uint16_t)xd_clip3((int)pred[yy * 64 + xx] + resi[yy * 64 + xx], 0, smax);
    }
  }
  if (pre_planes)
    for (int c = 0; c < 3; c++) {
      const int w = pic_w >> (c ? 1 : 0), h = pic_h >> (c ? 1 : 0);
      for (int y = 0; y < h; y++)
        memcpy(pre_planes[c] + (size_t)y * w, planes[c] + (ptrdiff_t)y * strides[c], 2 * w);
    }

  /* in-loop filter (picture_decoder.cc:186-191) */
  if (ps->deblock) {
    xvcgpu_cu_info *info = (xvcgpu_cu_info *)calloc((size_t)ps->n_cus, sizeof(*info));
    for (int i = 0; i < ps->n_cus; i++) {
      const xvc_cu_syntax *cu = &cus[i];
      xvcgpu_cu_info *o = &info[i];
      o->x = (uint16_t)cu->x;
      o->y = (uint16_t)cu->y;
      o->w = cu->w;
      o->h = cu->h;
      o->intra = cu->pred_mode == 0;
      o->cbf_luma = cu->cbf[0];
      o->qp_y = cu->qp[0];
      o->qp_c = cu->qp[1];
      o->ref_idx0 = cu->ref_idx[0];
      for (int l = 0; l < 2; l++) {
        const int used = cu->pred_mode == 1 && (cu->inter_dir == 2 || cu->inter_dir == l);
        o->ref_poc[l] = used ? ps->ref_poc[l][cu->ref_idx[l]] : -1;
        for (int k = 0; k < 3; k++) {
          const int src = (cu->flags & XVC_CU_AFFINE) ? k : 0;
          o->mv[l][k][0] = cu->mv[l][src][0];
          o->mv[l][k][1] = cu->mv[l][src][1];
        }
        /* CodingUnit::SetMv(MotionVector3) (coding_unit.h:268-275) */
        if (cu->flags & XVC_CU_AFFINE) {
          o->mv[l][3][0] = cu->mv[l][1][0] + cu->mv[l][2][0] - cu->mv[l][0][0];
          o->mv[l][3][1] = cu->mv[l][1][1] + cu->mv[l][2][1] - cu->mv[l][0][1];
        } else {
          o->mv[l][3][0] = cu->mv[l][0][0];
          o->mv[l][3][1] = cu->mv[l][0][1];
        }
      }
    }
    const int bipic = ps->pic_type == XVC_PIC_BI;
    if (two_trees) {
      xo_deblock_picture_planes(bd, pic_w, pic_h, bipic, ps->beta_offset, ps->tc_offset, 4, info,
                                m.cell[0], m.stride, planes, strides, 1);
      xo_deblock_picture_planes(bd, pic_w, pic_h, bipic, ps->beta_offset, ps->tc_offset, 8, info,
                                m.cell[1], m.stride, planes, strides, 2);
    } else {
      xo_deblock_picture_planes(bd, pic_w, pic_h, bipic, ps->beta_offset, ps->tc_offset, 4, info,
                                m.cell[0], m.stride, planes, strides, 3);
    }
    free(info);
  }
  if (ps->pad_border) {
    for (int c = 0; c < 3; c++) {
      const int s = c ? 1 : 0;
      xo_pad_border(pic_w >> s, pic_h >> s, border >> s, border >> s, planes[c], strides[c]);
    }
  }
  free(pred);
  free(m.cell[0]);
  free(m.cell[1]);
}
