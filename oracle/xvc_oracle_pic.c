/*
 * xvc_oracle_pic.c -- CPU restatement of the picture-level passes and the
 * motion-search control of the xvc hot path: deblocking, border extension,
 * TZ / full / sub-pel search, residual pipeline.
 *
 * TEST INFRASTRUCTURE ONLY (see xvc_oracle.h).  Parity status: pinned against
 * oracle/_ref and tests/golden/.
 */
#include <stdlib.h>
#include <string.h>

#include "xvc_oracle.h"

static inline int xp_clip3(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}
static inline uint16_t xp_clip_bd(int v, int max) {
  return (uint16_t)(v < 0 ? 0 : (v > max ? max : v));
}

/* ========================================================================= *
 *  Deblocking                                                               *
 * ========================================================================= */

/* kTcTable / kBetaTable: the HEVC-lineage normative deblocking tables
 * (deblocking_filter.cc:34-45). */
static const uint8_t xp_tc_table[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,  4,
    4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static const uint8_t xp_beta_table[64] = {
    0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
    6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56,
    58, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88};

typedef struct {
  int bd, pic_w, pic_h, bipred, beta_off, tc_off, sub;
  const xvcgpu_cu_info *cus;
  const int32_t *map;
  int map_stride, map_rows;
  int comp_mask; /* 1: filter luma, 2: filter chroma (DeblockCtu's deblock_luma /
                  * deblock_chroma, deblocking_filter.cc:88-91) */
  uint16_t *const *planes;
  const ptrdiff_t *strides;
} xp_db;

/* PictureData::GetCuAt (picture_data.h:102-107): the cell table has one extra
 * column/row beyond the CTU-aligned picture; outside the coded picture the
 * entry is nullptr.  Negative coordinates only occur as x-1 / y-1 at the
 * picture edge: the reference indexes cell (-1)/4 == 0 there for x = -1, i.e.
 * posx / 4 truncates toward zero, so (x-1) with x == 0 reads cell 0 of the
 * same row -> same CU -> "continue".  We return -1 (no CU) which also
 * continues. */
static const xvcgpu_cu_info *xp_cu_at(const xp_db *d, int x, int y) {
  if (x < 0 || y < 0) {
    /* mirror C truncating division: -1/4 == 0 */
    if (x < 0) x = 0;
    if (y < 0) y = 0;
  }
  int cx = x / 4, cy = y / 4;
  if (cx >= d->map_stride || cy >= d->map_rows) return NULL;
  int32_t idx = d->map[cy * d->map_stride + cx];
  return idx < 0 ? NULL : &d->cus[idx];
}

static int xp_abs(int v) { return v < 0 ? -v : v; }

/* GetBoundaryStrength, deblocking_filter.cc:154-241 (default restrictions) */
static int xp_bs(const xp_db *d, const xvcgpu_cu_info *p,
                 const xvcgpu_cu_info *q, int pos_x, int pos_y, int vertical) {
  const int one = 16; /* MotionVector::kScale */
  int cp, cq;
  if (vertical) {
    cp = (pos_y - p->y) < (p->h >> 1) ? XVC_CORNER_UR : XVC_CORNER_DR;
    cq = (pos_y - q->y) < (q->h >> 1) ? XVC_CORNER_UL : XVC_CORNER_DL;
  } else {
    cp = (pos_x - p->x) < (p->w >> 1) ? XVC_CORNER_DL : XVC_CORNER_DR;
    cq = (pos_x - q->x) < (q->w >> 1) ? XVC_CORNER_UL : XVC_CORNER_UR;
  }
  if (p->intra || q->intra) return 2;
  if (p->cbf_luma || q->cbf_luma) return 1;
  if (d->bipred) {
    int rp0 = p->ref_poc[0], rp1 = p->ref_poc[1];
    int rq0 = q->ref_poc[0], rq1 = q->ref_poc[1];
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
      const int32_t *p0 = p->mv[0][cp], *p1 = p->mv[1][cp];
      const int32_t *q0 = q->mv[0][cq], *q1 = q->mv[1][cq];
      int cond1 = xp_abs(p0[0] - q0[0]) >= one || xp_abs(p0[1] - q0[1]) >= one ||
                  xp_abs(p1[0] - q1[0]) >= one || xp_abs(p1[1] - q1[1]) >= one;
      int cond2 = xp_abs(p0[0] - q1[0]) >= one || xp_abs(p0[1] - q1[1]) >= one ||
                  xp_abs(p1[0] - q0[0]) >= one || xp_abs(p1[1] - q0[1]) >= one;
      if (rp0 != rp1) {
        if (rp0 == rq0) return cond1 ? 1 : 0;
        return cond2 ? 1 : 0;
      }
      return (cond1 && cond2) ? 1 : 0;
    }
    return 1;
  }
  if (p->ref_idx0 != q->ref_idx0) return 1;
  {
    const int32_t *p0 = p->mv[0][cp], *q0 = q->mv[0][cq];
    if (xp_abs(p0[0] - q0[0]) >= one || xp_abs(p0[1] - q0[1]) >= one) return 1;
  }
  return 0;
}

/* FilterEdgeLuma + weak/strong, deblocking_filter.cc:243-401 */
static void xp_filter_luma(const xp_db *d, int x, int y, int vertical, int bs,
                           int qp) {
  uint16_t *src = d->planes[0] + (ptrdiff_t)y * d->strides[0] + x;
  const ptrdiff_t stride = d->strides[0];
  const ptrdiff_t off = vertical ? 1 : stride;
  const ptrdiff_t step = vertical ? stride : 1;
  const int bsh = d->bd - 8;
  const int smax = (1 << d->bd) - 1;
  const int groups = d->sub / 4;
  for (int g = 0; g < groups; g++) {
    uint16_t *s = src + g * step * 4;
    int index_beta = xp_clip3(qp + d->beta_off, 0, 64);
    /* index 64 would be out of the table; qp + offset <= 63 in valid
     * streams */
    int beta = (index_beta < 64 ? xp_beta_table[index_beta] : 0) << bsh;
#define DP(ptr) xp_abs((int)(ptr)[-off * 3] - 2 * (int)(ptr)[-off * 2] + (int)(ptr)[-off])
#define DQ(ptr) xp_abs((int)(ptr)[0] - 2 * (int)(ptr)[off] + (int)(ptr)[off * 2])
    int dp0 = DP(s), dq0 = DQ(s);
    int dp3 = DP(s + step * 3), dq3 = DQ(s + step * 3);
#undef DP
#undef DQ
    int d0 = dp0 + dq0, d3 = dp3 + dq3, dd = d0 + d3;
    if (dd >= beta) continue;
    int index_tc = xp_clip3(qp + d->tc_off + 2 * (bs - 1), 0, 53);
    int tc = xp_tc_table[index_tc] << bsh;
    int strong = (d0 << 1) < (beta >> 2) && (d3 << 1) < (beta >> 2);
    for (int e = 0; e < 2 && strong; e++) { /* CheckStrongFilter :314-323 */
      const uint16_t *t = s + (e ? step * 3 : 0);
      int p3 = t[-off * 4], p0 = t[-off], q0 = t[0], q3 = t[off * 3];
      int test2 = (xp_abs(p3 - p0) + xp_abs(q0 - q3)) < (beta >> 3);
      int test3 = xp_abs(p0 - q0) < ((tc * 5 + 1) >> 1);
      strong = test2 && test3;
    }
    if (strong) { /* FilterLumaStrong :368-401 */
      const int tc2 = 2 * tc;
      uint16_t *t = s;
      for (int i = 0; i < 4; i++, t += step) {
        int p3 = t[-off * 4], p2 = t[-off * 3], p1 = t[-off * 2], p0 = t[-off];
        int q0 = t[0], q1 = t[off], q2 = t[off * 2], q3 = t[off * 3];
        int np2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        int np1 = (p2 + p1 + p0 + q0 + 2) >> 2;
        int np0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
        int nq0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
        int nq1 = (p0 + q0 + q1 + q2 + 2) >> 2;
        int nq2 = (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3;
        /* Sample + static_cast<Sample>(Clip3(...)) -> wraps mod 2^16 */
        t[-off * 3] = (uint16_t)(p2 + (uint16_t)xp_clip3(np2 - p2, -tc2, tc2));
        t[-off * 2] = (uint16_t)(p1 + (uint16_t)xp_clip3(np1 - p1, -tc2, tc2));
        t[-off] = (uint16_t)(p0 + (uint16_t)xp_clip3(np0 - p0, -tc2, tc2));
        t[0] = (uint16_t)(q0 + (uint16_t)xp_clip3(nq0 - q0, -tc2, tc2));
        t[off] = (uint16_t)(q1 + (uint16_t)xp_clip3(nq1 - q1, -tc2, tc2));
        t[off * 2] = (uint16_t)(q2 + (uint16_t)xp_clip3(nq2 - q2, -tc2, tc2));
      }
    } else { /* FilterLumaWeak :325-366 */
      int side = (beta + (beta >> 1)) >> 3;
      int filter_p1 = (dp0 + dp3) < side;
      int filter_q1 = (dq0 + dq3) < side;
      int threshold = tc * 10, half_tc = tc >> 1;
      uint16_t *t = s;
      for (int i = 0; i < 4; i++, t += step) {
        int p1 = t[-off * 2], p0 = t[-off], q0 = t[0], q1 = t[off];
        int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (xp_abs(delta) >= threshold) continue;
        delta = xp_clip3(delta, -tc, tc);
        t[-off] = xp_clip_bd(p0 + delta, smax);
        t[0] = xp_clip_bd(q0 - delta, smax);
        if (filter_p1) {
          int p2 = t[-off * 3];
          int dp1 = xp_clip3(((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1),
                             -half_tc, half_tc);
          t[-off * 2] = xp_clip_bd(p1 + dp1, smax);
        }
        if (filter_q1) {
          int q2 = t[off * 2];
          int dq1 = xp_clip3(((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1),
                             -half_tc, half_tc);
          t[off] = xp_clip_bd(q1 + dq1, smax);
        }
      }
    }
  }
}

/* FilterEdgeChroma / FilterChroma<N>, deblocking_filter.cc:403-450 (4:2:0) */
static void xp_filter_chroma(const xp_db *d, int cx, int cy, int vertical,
                             int qp) {
  const int bsh = d->bd - 8;
  const int smax = (1 << d->bd) - 1;
  int index_tc = xp_clip3(qp + d->tc_off + 2, 0, 54);
  int tc = (index_tc < 54 ? xp_tc_table[index_tc] : 0) << bsh;
  const int n = d->sub >> 1; /* scaled_subblock_size */
  for (int c = 1; c < 3; c++) {
    const ptrdiff_t stride = d->strides[c];
    uint16_t *t = d->planes[c] + (ptrdiff_t)cy * stride + cx;
    const ptrdiff_t off = vertical ? 1 : stride;
    const ptrdiff_t step = vertical ? stride : 1;
    for (int i = 0; i < n; i++, t += step) {
      int p1 = t[-off * 2], p0 = t[-off], q0 = t[0], q1 = t[off];
      int delta = xp_clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
      t[-off] = xp_clip_bd(p0 + delta, smax);
      t[0] = xp_clip_bd(q0 - delta, smax);
    }
  }
}

/* DeblockCtu, deblocking_filter.cc:79-152 (single CU tree, 4:2:0) */
static void xp_deblock_ctu(const xp_db *d, int ctu_x, int ctu_y, int vertical) {
  for (int dy = 0; dy < 64; dy += d->sub) {
    for (int dx = 0; dx < 64; dx += d->sub) {
      const int x = ctu_x + dx, y = ctu_y + dy;
      const xvcgpu_cu_info *q = xp_cu_at(d, x, y);
      if (!q) continue;
      const xvcgpu_cu_info *p =
          vertical ? xp_cu_at(d, x - 1, y) : xp_cu_at(d, x, y - 1);
      if (!p || (p->x == q->x && p->y == q->y)) continue;
      int bs = xp_bs(d, p, q, x, y, vertical);
      if (!bs) continue;
      int qp = (p->qp_y + q->qp_y + 1) >> 1;
      if (d->comp_mask & 1) xp_filter_luma(d, x, y, vertical, bs, qp);
      if (bs == 2 && (d->comp_mask & 2)) {
        int cqp = (p->qp_c + q->qp_c + 1) >> 1;
        int cx = x >> 1, cy = y >> 1;
        if (vertical ? ((cx & 7) == 0) : ((cy & 7) == 0))
          xp_filter_chroma(d, cx, cy, vertical, cqp);
      }
    }
  }
}

void xo_deblock_picture(int bd, int pic_w, int pic_h, int bipred,
                        int beta_offset, int tc_offset, int subblock_size,
                        const xvcgpu_cu_info *cus, const int32_t *cu_map,
                        int map_stride, uint16_t *const planes[3],
                        const ptrdiff_t strides[3]) {
  xo_deblock_picture_planes(bd, pic_w, pic_h, bipred, beta_offset, tc_offset, subblock_size,
                            cus, cu_map, map_stride, planes, strides, 3);
}

/* One CU tree's share of DeblockPicture (deblocking_filter.cc:56-77): with two
 * CU trees (intra pictures) the primary tree filters luma on the 4-sample
 * grid, the secondary tree chroma on the 8-sample grid; the two touch
 * disjoint planes, so the CTU interleaving of the reference is immaterial. */
void xo_deblock_picture_planes(int bd, int pic_w, int pic_h, int bipred,
                               int beta_offset, int tc_offset, int subblock_size,
                               const xvcgpu_cu_info *cus, const int32_t *cu_map,
                               int map_stride, uint16_t *const planes[3],
                               const ptrdiff_t strides[3], int comp_mask) {
  /* DeblockPicture, deblocking_filter.cc:56-77 */
  xp_db d;
  d.comp_mask = comp_mask;
  d.bd = bd;
  d.pic_w = pic_w;
  d.pic_h = pic_h;
  d.bipred = bipred;
  d.beta_off = beta_offset;
  d.tc_off = tc_offset;
  d.sub = subblock_size;
  d.cus = cus;
  d.map = cu_map;
  d.map_stride = map_stride;
  d.map_rows = (pic_h + 3) / 4;
  d.planes = planes;
  d.strides = strides;
  const int nx = (pic_w + 63) / 64, ny = (pic_h + 63) / 64;
  for (int vertical = 1; vertical >= 0; vertical--)
    for (int cy = 0; cy < ny; cy++)
      for (int cx = 0; cx < nx; cx++)
        xp_deblock_ctu(&d, cx * 64, cy * 64, vertical);
}

/* One pass restricted to subblock rows [y_begin, y_end): the shard unit of
 * the multi-GPU path (same semantics as xvcgpu_deblock_rows). */
void xo_deblock_rows(int bd, int pic_w, int pic_h, int bipred, int beta_offset,
                     int tc_offset, int subblock_size,
                     const xvcgpu_cu_info *cus, const int32_t *cu_map,
                     int map_stride, uint16_t *const planes[3],
                     const ptrdiff_t strides[3], int pass, int y_begin,
                     int y_end) {
  xp_db d;
  d.bd = bd;
  d.pic_w = pic_w;
  d.pic_h = pic_h;
  d.bipred = bipred;
  d.beta_off = beta_offset;
  d.tc_off = tc_offset;
  d.sub = subblock_size;
  d.cus = cus;
  d.map = cu_map;
  d.map_stride = map_stride;
  d.map_rows = (pic_h + 3) / 4;
  d.planes = planes;
  d.strides = strides;
  d.comp_mask = 3;
  const int vertical = pass == 0;
  if (y_end > pic_h) y_end = pic_h;
  for (int y = y_begin; y < y_end; y += subblock_size)
    for (int x = 0; x < pic_w; x += subblock_size) {
      const xvcgpu_cu_info *q = xp_cu_at(&d, x, y);
      if (!q) continue;
      const xvcgpu_cu_info *p =
          vertical ? xp_cu_at(&d, x - 1, y) : xp_cu_at(&d, x, y - 1);
      if (!p || (p->x == q->x && p->y == q->y)) continue;
      int bs = xp_bs(&d, p, q, x, y, vertical);
      if (!bs) continue;
      xp_filter_luma(&d, x, y, vertical, bs, (p->qp_y + q->qp_y + 1) >> 1);
      if (bs == 2) {
        int cx = x >> 1, cy = y >> 1;
        if (vertical ? ((cx & 7) == 0) : ((cy & 7) == 0))
          xp_filter_chroma(&d, cx, cy, vertical, (p->qp_c + q->qp_c + 1) >> 1);
      }
    }
}

/* ========================================================================= *
 *  Border extension                                                         *
 * ========================================================================= */

void xo_pad_border(int w, int h, int bx, int by, uint16_t *plane,
                   ptrdiff_t stride) {
  /* yuv_pic.cc:118-150 for one plane */
  uint16_t *row = plane;
  for (int y = -by; y < 0; y++)
    memcpy(row + y * stride, row, (size_t)w * sizeof(uint16_t));
  row += (ptrdiff_t)(h - 1) * stride;
  for (int y = 1; y <= by; y++)
    memcpy(row + y * stride, row, (size_t)w * sizeof(uint16_t));
  row = plane - (ptrdiff_t)by * stride;
  for (int y = 0; y < h + 2 * by; y++) {
    uint16_t left = row[0], right = row[w - 1];
    for (int x = -bx; x < 0; x++) row[x] = left;
    for (int x = 0; x < bx; x++) row[w + x] = right;
    row += stride;
  }
}

/* ========================================================================= *
 *  Motion search                                                            *
 * ========================================================================= */

static uint32_t xp_eg_bits(int mvd) { /* inter_search.cc:1179-1188 */
  uint32_t length = 1;
  uint32_t u = mvd <= 0 ? ((uint32_t)(-mvd) << 1) + 1 : ((uint32_t)mvd << 1);
  while (u != 1) {
    u >>= 1;
    length += 2;
  }
  return length;
}

uint32_t xo_mvd_bits_fullpel(int mvp_x, int mvp_y, int fx, int fy,
                             int mvd_down_shift) {
  /* inter_search.cc:1166-1177 */
  mvd_down_shift += 2;
  int mvd_x = ((fx * 16) - mvp_x) >> mvd_down_shift;
  int mvd_y = ((fy * 16) - mvp_y) >> mvd_down_shift;
  return xp_eg_bits(mvd_x) + xp_eg_bits(mvd_y);
}

uint32_t xo_mvd_bits(int mvp_x, int mvp_y, int mv_x, int mv_y,
                     int mvd_down_shift) {
  /* inter_search.cc:1150-1159 */
  int mvd_x = (mv_x - mvp_x) >> (2 + mvd_down_shift);
  int mvd_y = (mv_y - mvp_y) >> (2 + mvd_down_shift);
  return xp_eg_bits(mvd_x) + xp_eg_bits(mvd_y);
}

void xo_min_max_mv(int pos_x, int pos_y, int pic_w, int pic_h, int center_x,
                   int center_y, int search_range, int mv_min[2],
                   int mv_max[2]) {
  /* inter_prediction.cc:801-817; MvFullpel = mv >> 4 (cu_types.h:179-181) */
  xo_clip_mv(pos_x, pos_y, pic_w, pic_h, &center_x, &center_y);
  const int r = search_range << 4;
  int minx = center_x - r, miny = center_y - r;
  int maxx = center_x + r, maxy = center_y + r;
  xo_clip_mv(pos_x, pos_y, pic_w, pic_h, &minx, &miny);
  xo_clip_mv(pos_x, pos_y, pic_w, pic_h, &maxx, &maxy);
  mv_min[0] = minx >> 4;
  mv_min[1] = miny >> 4;
  mv_max[0] = maxx >> 4;
  mv_max[1] = maxy >> 4;
}

/* debug counters (tests/tools only): [0] SAD evaluations, [1] grid searches,
 * [2] refinement iterations, [3] searches */
__thread uint64_t xo_dbg_counters[4]; /* per thread: no cache-line ping-pong under OpenMP */

/* TZ search state, inter_tz_search.cc:66-82 */
typedef struct {
  int bd, w, h, metric;
  const uint16_t *orig, *ref; /* both at the CU position */
  ptrdiff_t os, rs;
  int mvp_x, mvp_y, down;
  int min[2], max[2];
  int best[2];
  uint64_t cost_best;
  int last_position, last_range;
  uint32_t lambda;
} xp_tz;

enum { XP_LEFT = -1, XP_RIGHT = 1, XP_UP = -3, XP_DOWN = 3 };

static int xp_check_best(xp_tz *s, int mx, int my) { /* :261-276 */
  const uint16_t *r = s->ref + (ptrdiff_t)my * s->rs + mx;
  xo_dbg_counters[0]++;
  uint64_t dist = xo_metric_ss(s->metric, s->bd, 0, 1, 1.0, s->w, s->h, s->orig,
                               s->os, r, s->rs);
  if (dist >= s->cost_best) return 0;
  uint32_t bits = xo_mvd_bits_fullpel(s->mvp_x, s->mvp_y, mx, my, s->down);
  uint64_t cost = dist + ((uint32_t)(s->lambda * bits) >> 16);
  if (cost < s->cost_best) {
    s->cost_best = cost;
    s->best[0] = mx;
    s->best[1] = my;
    return 1;
  }
  return 0;
}

static int xp_inside(const xp_tz *s, int dir, int mx, int my) { /* :278-302 */
  switch (dir) {
    case XP_UP: return my >= s->min[1];
    case XP_DOWN: return my <= s->max[1];
    case XP_LEFT: return mx >= s->min[0];
    default: return mx <= s->max[0];
  }
}

static int xp_check1(xp_tz *s, int dir, int mx, int my, int range) {
  if (!xp_inside(s, dir, mx, my)) return 0;
  if (!xp_check_best(s, mx, my)) return 0;
  s->last_position = dir;
  s->last_range = range;
  return 1;
}

static int xp_check2(xp_tz *s, int d1, int d2, int mx, int my, int range) {
  if (!xp_inside(s, d1, mx, my) || !xp_inside(s, d2, mx, my)) return 0;
  if (!xp_check_best(s, mx, my)) return 0;
  s->last_position = d1 + d2;
  s->last_range = range;
  return 1;
}

static int xp_diamond(xp_tz *s, int bx, int by, int range) { /* :173-210 */
  int mod = 0;
  if (range == 1) {
    mod |= xp_check1(s, XP_UP, bx, by - range, range);
    mod |= xp_check1(s, XP_LEFT, bx - range, by, range);
    mod |= xp_check1(s, XP_RIGHT, bx + range, by, range);
    mod |= xp_check1(s, XP_DOWN, bx, by + range, range);
  } else if (range <= 8) {
    int r2 = range >> 1;
    mod |= xp_check1(s, XP_UP, bx, by - range, range);
    mod |= xp_check2(s, XP_UP, XP_LEFT, bx - r2, by - r2, r2);
    mod |= xp_check2(s, XP_UP, XP_RIGHT, bx + r2, by - r2, r2);
    mod |= xp_check1(s, XP_LEFT, bx - range, by, range);
    mod |= xp_check1(s, XP_RIGHT, bx + range, by, range);
    mod |= xp_check2(s, XP_DOWN, XP_LEFT, bx - r2, by + r2, r2);
    mod |= xp_check2(s, XP_DOWN, XP_RIGHT, bx + r2, by + r2, r2);
    mod |= xp_check1(s, XP_DOWN, bx, by + range, range);
  } else {
    mod |= xp_check1(s, XP_UP, bx, by - range, range);
    mod |= xp_check1(s, XP_LEFT, bx - range, by, range);
    mod |= xp_check1(s, XP_RIGHT, bx + range, by, range);
    mod |= xp_check1(s, XP_DOWN, bx, by + range, range);
    for (int i = 1; i < 4; i++) {
      int r14 = i * (range >> 2), r34 = range - r14;
      mod |= xp_check2(s, XP_UP, XP_LEFT, bx - r14, by - r34, range);
      mod |= xp_check2(s, XP_UP, XP_RIGHT, bx + r14, by - r34, range);
      mod |= xp_check2(s, XP_DOWN, XP_LEFT, bx - r14, by + r34, range);
      mod |= xp_check2(s, XP_DOWN, XP_RIGHT, bx + r14, by + r34, range);
    }
  }
  return mod;
}

static void xp_neighbor(xp_tz *s) { /* FullpelNeighborPointSearch :212-259 */
  const int r = 1;
  int bx = s->best[0], by = s->best[1];
  switch (s->last_position) {
    case XP_UP + XP_LEFT:
      xp_check1(s, XP_LEFT, bx - r, by, r);
      xp_check1(s, XP_UP, bx, by - r, r);
      break;
    case XP_UP:
      xp_check2(s, XP_UP, XP_LEFT, bx - r, by - r, r);
      xp_check2(s, XP_UP, XP_RIGHT, bx + r, by - r, r);
      break;
    case XP_UP + XP_RIGHT:
      xp_check1(s, XP_UP, bx, by - r, r);
      xp_check1(s, XP_RIGHT, bx + r, by, r);
      break;
    case XP_LEFT:
      xp_check2(s, XP_DOWN, XP_LEFT, bx - r, by + r, r);
      xp_check2(s, XP_UP, XP_LEFT, bx - r, by - r, r);
      break;
    case XP_RIGHT:
      xp_check2(s, XP_UP, XP_RIGHT, bx + r, by - r, r);
      xp_check2(s, XP_DOWN, XP_RIGHT, bx + r, by + r, r);
      break;
    case XP_DOWN + XP_LEFT:
      xp_check1(s, XP_LEFT, bx - r, by, r);
      xp_check1(s, XP_DOWN, bx, by + r, r);
      break;
    case XP_DOWN:
      xp_check2(s, XP_DOWN, XP_LEFT, bx - r, by + r, r);
      xp_check2(s, XP_DOWN, XP_RIGHT, bx + r, by + r, r);
      break;
    case XP_DOWN + XP_RIGHT:
      xp_check1(s, XP_RIGHT, bx + r, by, r);
      xp_check1(s, XP_DOWN, bx, by + r, r);
      break;
    default:
      break;
  }
}

void xo_tz_search(int bd, const xvcgpu_me_block *b, int pic_w, int pic_h,
                  const uint16_t *orig, ptrdiff_t os, const uint16_t *ref,
                  ptrdiff_t rs, int out_mv[2], uint32_t *out_cost) {
  /* MotionEstNormal window (inter_search.cc:620-627) + TzSearch::Search
   * (inter_tz_search.cc:84-171) */
  xp_tz s;
  memset(&s, 0, sizeof(s));
  s.bd = bd;
  s.w = b->w;
  s.h = b->h;
  /* GetFullpelMetric (inter_search.cc:1059-1069): CUs that try local illumination
   * compensation compare with the block means removed */
  if (b->fullpel_mv & XVC_ME_USE_LIC)
    s.metric = b->h > 8 ? XVC_METRIC_SAD_ACONLY_FAST : XVC_METRIC_SAD_ACONLY;
  else
    s.metric = b->h > 8 ? XVC_METRIC_SAD_FAST : XVC_METRIC_SAD;
  s.orig = orig + (ptrdiff_t)b->y * os + b->x;
  s.ref = ref + (ptrdiff_t)b->y * rs + b->x;
  s.os = os;
  s.rs = rs;
  s.mvp_x = b->mvp_x;
  s.mvp_y = b->mvp_y;
  s.down = (b->fullpel_mv & XVC_ME_FULLPEL_MV) ? 2 : 0;
  s.lambda = b->lambda16;
  s.cost_best = UINT64_MAX;
  const int range = b->search_range;
  xo_min_max_mv(b->x, b->y, pic_w, pic_h, b->mvp_x, b->mvp_y, range, s.min,
                s.max);
  int fs_min[2] = {s.min[0], s.min[1]}, fs_max[2] = {s.max[0], s.max[1]};

  int cx = b->mvp_x, cy = b->mvp_y;
  xo_clip_mv(b->x, b->y, pic_w, pic_h, &cx, &cy);
  xp_check_best(&s, cx >> 4, cy >> 4);

  int change_min_max = 0;
  if (s.best[0] != 0 || s.best[1] != 0) change_min_max = xp_check_best(&s, 0, 0);
  s.last_range = 0;

  if (b->depth_nonzero) {
    int px = b->prev_x * 16, py = b->prev_y * 16;
    xo_clip_mv(b->x, b->y, pic_w, pic_h, &px, &py);
    change_min_max |= xp_check_best(&s, px >> 4, py >> 4);
    if (change_min_max)
      xo_min_max_mv(b->x, b->y, pic_w, pic_h, s.best[0] * 16, s.best[1] * 16,
                    range, fs_min, fs_max);
  }

  int base_x = s.best[0], base_y = s.best[1];
  int no_match = 0;
  for (int r = 1; r <= range; r *= 2) {
    if (xp_diamond(&s, base_x, base_y, r)) {
      no_match = 0;
    } else if (++no_match >= 3) {
      break;
    }
  }
  if (s.last_range == 1) {
    s.last_range = 0;
    xp_neighbor(&s);
  }
  xo_dbg_counters[3]++;
  if (s.last_range > 5) {
    xo_dbg_counters[1]++;
    s.last_range = 5;
    for (int y = fs_min[1]; y <= fs_max[1]; y += 5)
      for (int x = fs_min[0]; x <= fs_max[0]; x += 5) xp_check_best(&s, x, y);
  }
  while (s.last_range > 0) {
    xo_dbg_counters[2]++;
    int sx = s.best[0], sy = s.best[1];
    s.last_range = 0;
    for (int r = 1; r <= range; r *= 2) xp_diamond(&s, sx, sy, r);
    if (s.last_range == 1) {
      s.last_range = 0;
      xp_neighbor(&s);
    }
  }
  out_mv[0] = s.best[0];
  out_mv[1] = s.best[1];
  if (out_cost) *out_cost = (uint32_t)s.cost_best;
}

void xo_full_search(int bd, int x, int y, int w, int h, int fullpel_mv,
                    int mvp_x, int mvp_y, uint32_t lambda16, const int mv_min[2],
                    const int mv_max[2], const int16_t *target, ptrdiff_t ts,
                    const uint16_t *ref, ptrdiff_t rs, int out_mv[2]) {
  /* inter_search.cc:853-891 */
  const int metric = h > 8 ? XVC_METRIC_SAD_FAST : XVC_METRIC_SAD;
  const int down = fullpel_mv ? 2 : 0;
  const uint16_t *ref_cu = ref + (ptrdiff_t)y * rs + x;
  uint64_t cost_best = UINT64_MAX;
  int bx = 0, by = 0;
  for (int my = mv_min[1]; my <= mv_max[1]; my++)
    for (int mx = mv_min[0]; mx <= mv_max[0]; mx++) {
      uint64_t dist = xo_metric_rs(metric, bd, 0, 1, 1.0, w, h, target, ts,
                                   ref_cu + (ptrdiff_t)my * rs + mx, rs);
      if (dist >= cost_best) continue;
      uint32_t bits = xo_mvd_bits_fullpel(mvp_x, mvp_y, mx, my, down);
      uint64_t cost = dist + ((uint32_t)(lambda16 * bits) >> 16);
      if (cost < cost_best) {
        cost_best = cost;
        bx = mx;
        by = my;
      }
    }
  out_mv[0] = bx;
  out_mv[1] = by;
}

void xo_subpel_search(int bd, const xvcgpu_me_block *b, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os, const uint16_t *ref,
                      ptrdiff_t rs, const int fullpel[2], int out_mv[2],
                      uint32_t *out_dist) {
  /* inter_search.cc:893-964; offsets :38-43 */
  static const int8_t half[9][2] = {{0, 0},  {0, -1},  {0, 1},  {-1, 0}, {1, 0},
                                    {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
  static const int8_t qpel[9][2] = {{0, 0},  {0, -1}, {0, 1},  {-1, -1}, {1, -1},
                                    {-1, 0}, {1, 0},  {-1, 1}, {1, 1}};
  uint16_t pred[64 * 64];
  const uint16_t *o = orig + (ptrdiff_t)b->y * os + b->x;
  uint64_t best_cost = UINT64_MAX, best_dist = UINT64_MAX;
  int best_x = fullpel[0] * 16, best_y = fullpel[1] * 16;
  for (int pass = 0; pass < 2; pass++) {
    const int base_x = best_x, base_y = best_y;
    const int scale = pass == 0 ? 8 : 4; /* MvDelta(.,.,1|2) -> 1/16 units */
    for (int i = pass; i < 9; i++) {
      const int8_t *d = pass == 0 ? half[i] : qpel[i];
      const int mx = base_x + d[0] * scale, my = base_y + d[1] * scale;
      xo_mc_block(bd, 0, b->x, b->y, b->w, b->h, mx, my, pic_w, pic_h, ref, rs,
                  pred, 64);
      /* GetSubpelMetric (inter_search.cc:1071-1076) */
      uint64_t dist = xo_metric_ss((b->fullpel_mv & XVC_ME_USE_LIC) ? XVC_METRIC_SATD_ACONLY
                                                                    : XVC_METRIC_SATD,
                                   bd, 0, 1, 1.0, b->w, b->h, o, os, pred, 64);
      if (dist >= best_cost) continue;
      uint32_t bits = xo_mvd_bits(b->mvp_x, b->mvp_y, mx, my, 0);
      uint64_t cost = dist + ((uint32_t)(b->lambda16 * bits) >> 16);
      if (cost < best_cost) {
        best_cost = cost;
        best_dist = dist;
        best_x = mx;
        best_y = my;
      }
    }
  }
  out_mv[0] = best_x;
  out_mv[1] = best_y;
  if (out_dist) *out_dist = (uint32_t)best_dist;
}

uint64_t xo_mc_metric(int bd, int metric, int qp_raw, int strength, int x, int y, int w,
                      int h, int mv_x, int mv_y, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os, const uint16_t *ref,
                      ptrdiff_t rs) {
  /* GetSubpelDist, inter_search.cc:951-964 (the per-candidate step of
   * EvalStartMvp :966-997 and SearchMergeCandidates :165-197) */
  uint16_t pred[64 * 64];
  xo_mc_block(bd, 0, x, y, w, h, mv_x, mv_y, pic_w, pic_h, ref, rs, pred, 64);
  return xo_metric_ss(metric, bd, qp_raw, strength, 1.0, w, h,
                      orig + (ptrdiff_t)y * os + x, os, pred, 64);
}

/* SubpelSearch on an int16 target (bi-pred: TOrig = Residual), same fold. */
static void xp_subpel_search_rs(int bd, const xvcgpu_me_block *b, int pic_w,
                                int pic_h, const int16_t *target, ptrdiff_t ts,
                                const uint16_t *ref, ptrdiff_t rs,
                                const int fullpel[2], int out_mv[2],
                                uint32_t *out_dist) {
  static const int8_t half[9][2] = {{0, 0},  {0, -1},  {0, 1},  {-1, 0}, {1, 0},
                                    {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
  static const int8_t qpel[9][2] = {{0, 0},  {0, -1}, {0, 1},  {-1, -1}, {1, -1},
                                    {-1, 0}, {1, 0},  {-1, 1}, {1, 1}};
  uint16_t pred[64 * 64];
  uint64_t best_cost = UINT64_MAX, best_dist = UINT64_MAX;
  int best_x = fullpel[0] * 16, best_y = fullpel[1] * 16;
  for (int pass = 0; pass < 2; pass++) {
    const int base_x = best_x, base_y = best_y;
    const int scale = pass == 0 ? 8 : 4;
    for (int i = pass; i < 9; i++) {
      const int8_t *d = pass == 0 ? half[i] : qpel[i];
      const int mx = base_x + d[0] * scale, my = base_y + d[1] * scale;
      xo_mc_block(bd, 0, b->x, b->y, b->w, b->h, mx, my, pic_w, pic_h, ref, rs,
                  pred, 64);
      uint64_t dist = xo_metric_rs(XVC_METRIC_SATD, bd, 0, 1, 1.0, b->w, b->h,
                                   target, ts, pred, 64);
      if (dist >= best_cost) continue;
      uint32_t bits = xo_mvd_bits(b->mvp_x, b->mvp_y, mx, my, 0);
      uint64_t cost = dist + ((uint32_t)(b->lambda16 * bits) >> 16);
      if (cost < best_cost) {
        best_cost = cost;
        best_dist = dist;
        best_x = mx;
        best_y = my;
      }
    }
  }
  out_mv[0] = best_x;
  out_mv[1] = best_y;
  *out_dist = (uint32_t)best_dist;
}

void xo_bipred_search(int bd, const xvcgpu_bi_block *j, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t os,
                      const uint16_t *ref_other, ptrdiff_t ros,
                      const uint16_t *ref_search, ptrdiff_t rss,
                      xvcgpu_me_result *out) {
  /* One inner step of SearchBiIterative (inter_search.cc:392-433) +
   * MotionEstNormal(kFullSearch, bipred) (:606-662) */
  const xvcgpu_me_block *b = &j->blk;
  uint16_t pred[64 * 64];
  int16_t target[64 * 64];
  xo_mc_block(bd, 0, b->x, b->y, b->w, b->h, j->other_mv_x, j->other_mv_y, pic_w,
              pic_h, ref_other, ros, pred, 64);
  for (int y = 0; y < b->h; y++) /* SubtractWeighted, sample_buffer.h:147-161 */
    for (int x = 0; x < b->w; x++)
      target[y * 64 + x] =
          (int16_t)(2 * (int)orig[(ptrdiff_t)(b->y + y) * os + b->x + x] -
                    (int)pred[y * 64 + x]);
  int mn[2], mx[2], fp[2], mv[2];
  xo_min_max_mv(b->x, b->y, pic_w, pic_h, j->boot_mv_x, j->boot_mv_y, 4, mn, mx);
  xo_full_search(bd, b->x, b->y, b->w, b->h, b->fullpel_mv, b->mvp_x, b->mvp_y,
                 b->lambda16, mn, mx, target, 64, ref_search, rss, fp);
  uint32_t dist = 0;
  if (b->fullpel_mv) {
    mv[0] = fp[0] * 16;
    mv[1] = fp[1] * 16;
    xo_mc_block(bd, 0, b->x, b->y, b->w, b->h, mv[0], mv[1], pic_w, pic_h,
                ref_search, rss, pred, 64);
    dist = (uint32_t)xo_metric_rs(XVC_METRIC_SATD, bd, 0, 1, 1.0, b->w, b->h,
                                  target, 64, pred, 64);
  } else {
    xp_subpel_search_rs(bd, b, pic_w, pic_h, target, 64, ref_search, rss, fp, mv,
                        &dist);
  }
  out->fullpel_x = fp[0];
  out->fullpel_y = fp[1];
  out->mv_x = mv[0];
  out->mv_y = mv[1];
  out->fullpel_cost = 0;
  out->subpel_dist = dist >> 1; /* *out_dist = bipred ? dist >> 1 : dist */
}

void xo_mc_bipred_block(int bd, int comp, int x, int y, int w, int h, int mv0_x,
                        int mv0_y, int mv1_x, int mv1_y, int pic_w, int pic_h,
                        const uint16_t *ref0, ptrdiff_t rs0, const uint16_t *ref1,
                        ptrdiff_t rs1, uint16_t *pred, ptrdiff_t ps) {
  /* MotionCompensation, normal bi-prediction (inter_prediction.cc:710-738) */
  int16_t p0[64 * 64], p1[64 * 64];
  const int cs = comp ? 1 : 0, shift = 4 + cs;
  const int cx = x >> cs, cy = y >> cs, cw = w >> cs, ch = h >> cs;
  const uint16_t *refs[2] = {ref0, ref1};
  const ptrdiff_t rss[2] = {rs0, rs1};
  int mvs[2][2] = {{mv0_x, mv0_y}, {mv1_x, mv1_y}};
  int16_t *outs[2] = {p0, p1};
  for (int l = 0; l < 2; l++) {
    int mx = mvs[l][0], my = mvs[l][1];
    xo_clip_mv(x, y, pic_w, pic_h, &mx, &my);
    const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
    const uint16_t *r = refs[l] + (ptrdiff_t)(cy + (my >> shift)) * rss[l] + cx + (mx >> shift);
    xo_mc_uni_bipred(bd, comp != 0, cw, ch, fx, fy, r, rss[l], outs[l], 64);
  }
  xo_add_avg(bd, cw, ch, p0, 64, p1, 64, pred, ps);
}

/* ========================================================================= *
 *  Residual pipeline                                                        *
 * ========================================================================= */

int xo_residual_pipeline(int bd, const xvcgpu_tx_block *b, const uint16_t *orig,
                         ptrdiff_t os, const uint16_t *pred, ptrdiff_t ps,
                         uint16_t *rec, ptrdiff_t rs, int16_t *coeff_out) {
  /* transform_encoder.cc:203-285 with QuantFast, no transform skip */
  int16_t resi[64 * 64], coeff[64 * 64], deq[64 * 64];
  const int w = b->w, h = b->h;
  memset(resi, 0, sizeof(resi));
  const uint16_t *o = orig + (ptrdiff_t)b->y * os + b->x;
  const uint16_t *p = pred + (ptrdiff_t)b->y * ps + b->x;
  uint16_t *r = rec + (ptrdiff_t)b->y * rs + b->x;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      resi[y * 64 + x] = (int16_t)((int)o[y * os + x] - (int)p[y * ps + x]);
  const int skip = b->tx_hor == XVC_TX_SKIP; /* cu->GetTransformSkip(comp) */
  if (skip)
    xo_fwd_transform_skip(bd, w, h, resi, 64, coeff, 64);
  else
    xo_fwd_transform(bd, w, h, b->tx_hor, b->tx_ver, b->dst4x4, resi, 64, coeff, 64);
  int nnz = xo_quant_fast2(bd, b->qp, b->intra_pic & XVC_TXF_INTRA_PIC,
                           !(b->intra_pic & XVC_TXF_NO_SIGN_HIDING),
                           (b->intra_pic >> XVC_TXF_SCAN_SHIFT) & 3, w, h, coeff, 64,
                           coeff_out, w);
  if (nnz) {
    int dc_only = nnz == 1 && coeff_out[0] != 0;
    xo_dequant(bd, b->qp, w, h, coeff_out, w, deq, 64);
    if (skip)
      xo_inv_transform_skip(bd, w, h, deq, 64, resi, 64);
    else
      xo_inv_transform(bd, w, h, b->tx_hor, b->tx_ver, b->dst4x4, dc_only, deq, 64,
                       resi, 64);
    const int smax = (1 << bd) - 1;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        r[y * rs + x] = (uint16_t)xp_clip3((int)p[y * ps + x] + resi[y * 64 + x],
                                           0, smax);
  } else {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) r[y * rs + x] = p[y * ps + x];
  }
  return nnz;
}

/* =========================================================================
 *  Local illumination compensation
 * ========================================================================= */

/* MotionCompensationMv(post_filter = true) of a CU with use_lic
 * (inter_prediction.cc:740-758): the ordinary uni-pred prediction, then
 * LocalIlluminationComp (:1555-1575) with the model of DeriveLicParams
 * (:1577-1663).  `ref` = padded reference plane of the component, `rec` = the
 * current reconstruction plane (neighbours of the block must be there); both
 * pointers at sample (0,0). */
void xo_mc_lic_block(int bitdepth, const xvcgpu_mc_lic_block *b, int pic_w, int pic_h,
                     const uint16_t *ref, ptrdiff_t rs, const uint16_t *rec, ptrdiff_t cs,
                     uint16_t *pred, ptrdiff_t ps) {
  const int c = b->comp, sub = c ? 1 : 0;
  xo_mc_block(bitdepth, c, b->x, b->y, b->w, b->h, b->mv_x, b->mv_y, pic_w, pic_h, ref, rs,
              pred + (ptrdiff_t)(b->y >> sub) * ps + (b->x >> sub), ps);
  int mx = b->mv_x, my = b->mv_y;
  xo_clip_mv(b->x, b->y, pic_w, pic_h, &mx, &my);
  const int shift = 4 + sub;
  const int fx = (mx + (1 << (shift - 1))) >> shift, fy = (my + (1 << (shift - 1))) >> shift;
  const int w = b->w >> sub, h = b->h >> sub, x = b->x >> sub, y = b->y >> sub;
  const int has_above = b->neighbors & XVC_LIC_HAS_ABOVE, has_left = b->neighbors & XVC_LIC_HAS_LEFT;
  int scale = 32, offset = 0;
  if (has_above || has_left) {
    const int step = (w < h ? w : h) > 8 ? 2 : 1;
    const uint16_t *rb = ref + (ptrdiff_t)y * rs + x, *sb = rec + (ptrdiff_t)y * cs + x;
    int sum_x = 0, sum_y = 0, sum_xx = 0, sum_xy = 0, nbr = 0;
    if (has_above) {
      int cx = fx, cy = fy; /* the full-pel vector goes through ClipMv as it is */
      xo_clip_mv(b->above_x, b->above_y, pic_w, pic_h, &cx, &cy);
      const uint16_t *r = rb + cx + (ptrdiff_t)cy * rs - rs, *s2 = sb - cs;
      const int dx = step * (w / h > 1 ? w / h : 1);
      for (int i = 0; i < w; i += dx) {
        sum_x += r[i]; sum_y += s2[i]; sum_xx += r[i] * r[i]; sum_xy += r[i] * s2[i]; nbr++;
      }
    }
    if (has_left) {
      int cx = fx, cy = fy;
      xo_clip_mv(b->left_x, b->left_y, pic_w, pic_h, &cx, &cy);
      const uint16_t *r = rb + cx + (ptrdiff_t)cy * rs - 1, *s2 = sb - 1;
      const int dy = step * (h / w > 1 ? h / w : 1);
      for (int i = 0; i < h; i += dy) {
        const int a = r[i * rs], d = s2[i * cs];
        sum_x += a; sum_y += d; sum_xx += a * a; sum_xy += a * d; nbr++;
      }
    }
    int size_shift = 1; /* util::SizeToLog2 */
    while ((1 << size_shift) < nbr) size_shift++;
    int base_shift = bitdepth + size_shift - 15;
    if (base_shift < 0) base_shift = 0;
    const int avg_x = sum_x >> base_shift, avg_y = sum_y >> base_shift;
    const int xx_offset = sum_xx >> 7;
    const int avg_xy = ((sum_xy + xx_offset) >> (2 * base_shift)) << size_shift;
    const int avg_xx = ((sum_xx + xx_offset) >> (2 * base_shift)) << size_shift;
    const int sxy = avg_xy - avg_x * avg_y, sxx = avg_xx - avg_x * avg_x;
    int msb = 0;
    for (unsigned v = (unsigned)abs(sxx); v; v >>= 1) msb++;
    int shift_xx = msb - 6;
    if (shift_xx < 0) shift_xx = 0;
    int shift_xy = shift_xx - 12;
    if (shift_xy < 0) shift_xy = 0;
    const int total_shift = 15 - 5 + shift_xx - shift_xy;
    const int sxy_s = sxy >> shift_xy;
    int sxx_s = sxx >> shift_xx;
    sxx_s = sxx_s < 0 ? 0 : (sxx_s > 63 ? 63 : sxx_s);
    if (sxx_s != 0) {
      const int sxx_scaled = ((1 << 15) + (sxx_s / 2)) / sxx_s;
      int sc = (int)((int64_t)sxy_s * sxx_scaled) >> total_shift;
      scale = sc < 0 ? 0 : (sc > 128 ? 128 : sc);
      int off = (sum_y - ((scale * sum_x) >> 5) + (1 << (size_shift - 1))) >> size_shift;
      const int lo = -(1 << (bitdepth - 1)), hi = (1 << (bitdepth - 1)) - 1;
      offset = off < lo ? lo : (off > hi ? hi : off);
    }
  }
  const int max = (1 << bitdepth) - 1;
  uint16_t *p = pred + (ptrdiff_t)y * ps + x;
  for (int yy = 0; yy < h; yy++)
    for (int xx = 0; xx < w; xx++) {
      const int v = ((scale * p[yy * ps + xx]) >> 5) + offset;
      p[yy * ps + xx] = (uint16_t)(v < 0 ? 0 : (v > max ? max : v));
    }
}
