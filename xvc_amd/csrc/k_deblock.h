// k_deblock.h -- D1..D4: DeblockingFilter::DeblockPicture
// (deblocking_filter.cc:56-450) as two whole-picture passes.
//
// Parallel decomposition: an edge group filters 4
// lines and touches samples p3..q3 across the edge (writes p2..q2).  Within a
// pass, bands of 4 lines are independent; along a band two edges interact
// only when they are exactly 4 samples apart (both exist only around 4-wide /
// 4-tall CUs).  Whether an edge position is a *candidate* (different CUs on
// both sides and boundary strength > 0) depends only on the CU map, never on
// samples.  So: one thread per (band, subblock position); a thread whose
// predecessor position is also a candidate does nothing; a chain head filters
// its edge and then walks the chain in increasing x (pass 1) / y (pass 2),
// which is exactly the reference's raster order restricted to that band.
// With the 8-sample grid every candidate is its own chain.
//
// Memory: lanes of a wave map to consecutive positions ALONG the edge
// direction's orthogonal axis that is contiguous in memory where possible:
//   pass 1 (vertical edges):   lane -> x position, each line = 16 contiguous B
//   pass 2 (horizontal edges): lane -> x position, each row  =  8 contiguous B
// so both passes issue row-contiguous 8/16-byte accesses.
#ifndef XVCGPU_K_DEBLOCK_H_
#define XVCGPU_K_DEBLOCK_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "xvcgpu_internal.h"

struct DbParams {
  int bd, pic_w, pic_h, bipred, beta_off, tc_off, sub;
  int y_begin, y_end;  // subblock rows [y_begin, y_end) handled by this launch
  const xvcgpu_cu_info *cus;
  const int32_t *map;
  int map_stride, map_rows;
  int comp_mask;  // 1: filter luma, 2: filter chroma (DeblockCtu's deblock_luma /
                  // deblock_chroma, deblocking_filter.cc:88-91)
};

__device__ __forceinline__ int db_cu_index(const DbParams &d, int x, int y) {
  // PictureData::GetCuAt with C truncating division for x-1 / y-1 == -1
  if (x < 0) x = 0;
  if (y < 0) y = 0;
  const int cx = x >> 2, cy = y >> 2;
  if (cx >= d.map_stride || cy >= d.map_rows) return -1;
  return d.map[cy * d.map_stride + cx];
}

// GetBoundaryStrength, deblocking_filter.cc:154-241 (default restrictions).
__device__ __forceinline__ int db_bs(const DbParams &d,
                                     const xvcgpu_cu_info &p,
                                     const xvcgpu_cu_info &q, int pos_x,
                                     int pos_y, bool vertical) {
  const int one = 16;
  int cp, cq;
  if (vertical) {
    cp = (pos_y - p.y) < (p.h >> 1) ? XVC_CORNER_UR : XVC_CORNER_DR;
    cq = (pos_y - q.y) < (q.h >> 1) ? XVC_CORNER_UL : XVC_CORNER_DL;
  } else {
    cp = (pos_x - p.x) < (p.w >> 1) ? XVC_CORNER_DL : XVC_CORNER_DR;
    cq = (pos_x - q.x) < (q.w >> 1) ? XVC_CORNER_UL : XVC_CORNER_UR;
  }
  if (p.intra || q.intra) return 2;
  if (p.cbf_luma || q.cbf_luma) return 1;
  if (d.bipred) {
    const int rp0 = p.ref_poc[0], rp1 = p.ref_poc[1];
    const int rq0 = q.ref_poc[0], rq1 = q.ref_poc[1];
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
      const int32_t *p0 = p.mv[0][cp], *p1 = p.mv[1][cp];
      const int32_t *q0 = q.mv[0][cq], *q1 = q.mv[1][cq];
      const bool cond1 =
          d_abs(p0[0] - q0[0]) >= one || d_abs(p0[1] - q0[1]) >= one ||
          d_abs(p1[0] - q1[0]) >= one || d_abs(p1[1] - q1[1]) >= one;
      const bool cond2 =
          d_abs(p0[0] - q1[0]) >= one || d_abs(p0[1] - q1[1]) >= one ||
          d_abs(p1[0] - q0[0]) >= one || d_abs(p1[1] - q0[1]) >= one;
      if (rp0 != rp1) return (rp0 == rq0) ? (cond1 ? 1 : 0) : (cond2 ? 1 : 0);
      return (cond1 && cond2) ? 1 : 0;
    }
    return 1;
  }
  if (p.ref_idx0 != q.ref_idx0) return 1;
  const int32_t *p0 = p.mv[0][cp], *q0 = q.mv[0][cq];
  return (d_abs(p0[0] - q0[0]) >= one || d_abs(p0[1] - q0[1]) >= one) ? 1 : 0;
}

// Candidate test for the subblock position (x,y); returns bs (0 = none) and
// the luma / chroma qp averages (deblocking_filter.cc:98-135).
__device__ __forceinline__ int db_candidate(const DbParams &d, int x, int y,
                                            bool vertical, int &qp, int &cqp) {
  if (x >= d.pic_w || y >= d.pic_h || x < 0 || y < 0) return 0;
  const int iq = db_cu_index(d, x, y);
  if (iq < 0) return 0;
  const int ip = vertical ? db_cu_index(d, x - 1, y) : db_cu_index(d, x, y - 1);
  if (ip < 0 || ip == iq) return 0;
  const xvcgpu_cu_info &p = d.cus[ip], &q = d.cus[iq];
  if (p.x == q.x && p.y == q.y) return 0;
  const int bs = db_bs(d, p, q, x, y, vertical);
  qp = (p.qp_y + q.qp_y + 1) >> 1;
  cqp = (p.qp_c + q.qp_c + 1) >> 1;
  return bs;
}

// One 4-line group of FilterEdgeLuma (deblocking_filter.cc:243-401).
// s[line][0..7] = p3,p2,p1,p0,q0,q1,q2,q3.  Returns true when modified.
// (beta, tc: the table values already scaled to the bit depth)
__device__ __forceinline__ bool db_filter_luma_group_bt(int s[4][8], int bd, int beta, int tc) {
  const int smax = (1 << bd) - 1;
  const int dp0 = d_abs(s[0][1] - 2 * s[0][2] + s[0][3]);
  const int dq0 = d_abs(s[0][4] - 2 * s[0][5] + s[0][6]);
  const int dp3 = d_abs(s[3][1] - 2 * s[3][2] + s[3][3]);
  const int dq3 = d_abs(s[3][4] - 2 * s[3][5] + s[3][6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3;
  if (d0 + d3 >= beta) return false;
  bool strong = (d0 << 1) < (beta >> 2) && (d3 << 1) < (beta >> 2);
#pragma unroll
  for (int e = 0; e < 4; e += 3) {  // CheckStrongFilter on lines 0 and 3
    const int p3 = s[e][0], p0 = s[e][3], q0 = s[e][4], q3 = s[e][7];
    strong = strong && (d_abs(p3 - p0) + d_abs(q0 - q3)) < (beta >> 3) &&
             d_abs(p0 - q0) < ((tc * 5 + 1) >> 1);
  }
  if (strong) {
    const int tc2 = 2 * tc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int p3 = s[i][0], p2 = s[i][1], p1 = s[i][2], p0 = s[i][3];
      const int q0 = s[i][4], q1 = s[i][5], q2 = s[i][6], q3 = s[i][7];
      const int np2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
      const int np1 = (p2 + p1 + p0 + q0 + 2) >> 2;
      const int np0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
      const int nq0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
      const int nq1 = (p0 + q0 + q1 + q2 + 2) >> 2;
      const int nq2 = (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3;
      // Sample + static_cast<Sample>(Clip3(..)) wraps modulo 2^16
      s[i][1] = (p2 + d_clip3(np2 - p2, -tc2, tc2)) & 0xffff;
      s[i][2] = (p1 + d_clip3(np1 - p1, -tc2, tc2)) & 0xffff;
      s[i][3] = (p0 + d_clip3(np0 - p0, -tc2, tc2)) & 0xffff;
      s[i][4] = (q0 + d_clip3(nq0 - q0, -tc2, tc2)) & 0xffff;
      s[i][5] = (q1 + d_clip3(nq1 - q1, -tc2, tc2)) & 0xffff;
      s[i][6] = (q2 + d_clip3(nq2 - q2, -tc2, tc2)) & 0xffff;
    }
    return true;
  }
  const int side = (beta + (beta >> 1)) >> 3;
  const bool filter_p1 = (dp0 + dp3) < side, filter_q1 = (dq0 + dq3) < side;
  const int threshold = tc * 10, half_tc = tc >> 1;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p2 = s[i][1], p1 = s[i][2], p0 = s[i][3];
    const int q0 = s[i][4], q1 = s[i][5], q2 = s[i][6];
    int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
    if (d_abs(delta) >= threshold) continue;
    delta = d_clip3(delta, -tc, tc);
    s[i][3] = d_clip_bd(p0 + delta, smax);
    s[i][4] = d_clip_bd(q0 - delta, smax);
    if (filter_p1) {
      const int dp1 =
          d_clip3(((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1), -half_tc, half_tc);
      s[i][2] = d_clip_bd(p1 + dp1, smax);
    }
    if (filter_q1) {
      const int dq1 =
          d_clip3(((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1), -half_tc, half_tc);
      s[i][5] = d_clip_bd(q1 + dq1, smax);
    }
  }
  return true;
}

__device__ __forceinline__ int db_beta_index(int qp, int beta_off) {
  return d_clip3(qp + beta_off, 0, 64);   // 64: beta = 0
}
__device__ __forceinline__ int db_tc_index(int qp, int tc_off, int bs) {
  return d_clip3(qp + tc_off + 2 * (bs - 1), 0, 53);
}

__device__ __forceinline__ bool db_filter_luma_group(int s[4][8], int bd, int qp,
                                                     int bs, int beta_off,
                                                     int tc_off) {
  const int bsh = bd - 8;
  const int index_beta = db_beta_index(qp, beta_off);
  const int beta = (index_beta < 64 ? (int)kBetaTable[index_beta] : 0) << bsh;
  const int tc = (int)kTcTable[db_tc_index(qp, tc_off, bs)] << bsh;
  return db_filter_luma_group_bt(s, bd, beta, tc);
}

__device__ __forceinline__ void db_unpack8(const uint2 a, const uint2 b, int *o) {
  o[0] = a.x & 0xffff; o[1] = a.x >> 16; o[2] = a.y & 0xffff; o[3] = a.y >> 16;
  o[4] = b.x & 0xffff; o[5] = b.x >> 16; o[6] = b.y & 0xffff; o[7] = b.y >> 16;
}
__device__ __forceinline__ uint32_t db_pack(int lo, int hi) {
  return (uint32_t)lo | ((uint32_t)hi << 16);
}

// Filter the luma edge at (x,y) (all `sub/4` groups) and, for bs == 2, the
// chroma edges (FilterEdgeChroma, deblocking_filter.cc:403-450).
template <bool VERTICAL>
__device__ __forceinline__ void db_filter_edge(const DbParams &d,
                                               const PicView &pic, int x, int y,
                                               int bs, int qp, int cqp) {
  const PlaneView pl = pic.c[0];
  for (int g = 0; g < ((d.comp_mask & 1) ? d.sub / 4 : 0); g++) {
    int s[4][8];
    if (VERTICAL) {
      uint16_t *base = pl.p + (ptrdiff_t)(y + 4 * g) * pl.stride + x - 4;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint2 *r = reinterpret_cast<const uint2 *>(base + (ptrdiff_t)i * pl.stride);
        db_unpack8(r[0], r[1], s[i]);
      }
      if (db_filter_luma_group(s, d.bd, qp, bs, d.beta_off, d.tc_off)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          uint2 *r = reinterpret_cast<uint2 *>(base + (ptrdiff_t)i * pl.stride);
          r[0] = make_uint2(db_pack(s[i][0], s[i][1]), db_pack(s[i][2], s[i][3]));
          r[1] = make_uint2(db_pack(s[i][4], s[i][5]), db_pack(s[i][6], s[i][7]));
        }
      }
    } else {
      uint16_t *base = pl.p + (ptrdiff_t)(y - 4) * pl.stride + x + 4 * g;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint2 r = *reinterpret_cast<const uint2 *>(base + (ptrdiff_t)k * pl.stride);
        s[0][k] = r.x & 0xffff;
        s[1][k] = r.x >> 16;
        s[2][k] = r.y & 0xffff;
        s[3][k] = r.y >> 16;
      }
      if (db_filter_luma_group(s, d.bd, qp, bs, d.beta_off, d.tc_off)) {
#pragma unroll
        for (int k = 1; k < 7; k++)
          *reinterpret_cast<uint2 *>(base + (ptrdiff_t)k * pl.stride) =
              make_uint2(db_pack(s[0][k], s[1][k]), db_pack(s[2][k], s[3][k]));
      }
    }
  }
  if (bs != 2 || !(d.comp_mask & 2)) return;
  const int cx = x >> 1, cy = y >> 1;
  if (VERTICAL ? (cx & 7) != 0 : (cy & 7) != 0) return;
  const int bsh = d.bd - 8, smax = (1 << d.bd) - 1;
  const int index_tc = d_clip3(cqp + d.tc_off + 2, 0, 54);
  const int tc = (index_tc < 54 ? (int)kTcTable[index_tc] : 0) << bsh;
  const int n = d.sub >> 1;
  for (int c = 1; c < 3; c++) {
    const PlaneView pc = pic.c[c];
    for (int i = 0; i < n; i++) {
      uint16_t *t = pc.p + (ptrdiff_t)(cy + (VERTICAL ? i : 0)) * pc.stride + cx +
                    (VERTICAL ? 0 : i);
      const ptrdiff_t off = VERTICAL ? 1 : pc.stride;
      const int p1 = t[-off * 2], p0 = t[-off], q0 = t[0], q1 = t[off];
      const int delta = d_clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
      t[-off] = d_clip_bd(p0 + delta, smax);
      t[0] = d_clip_bd(q0 - delta, smax);
    }
  }
}

// grid: (ceil(nx/64), ny) with nx = ceil(pic_w/sub), ny = ceil(pic_h/sub);
// block: 64.  Thread = subblock position (x = sub*ix, y = sub*iy).
template <bool VERTICAL>
__global__ void __launch_bounds__(64)
deblock_pass_kernel(DbParams d, PicView pic) {
  const int ix = blockIdx.x * 64 + threadIdx.x, iy = blockIdx.y;
  int x = ix * d.sub, y = d.y_begin + iy * d.sub;
  if (x >= d.pic_w || y >= d.pic_h || y >= d.y_end) return;
  int qp, cqp;
  int bs = db_candidate(d, x, y, VERTICAL, qp, cqp);
  if (!bs) return;
  if (d.sub == 4) {
    // not a chain head if the previous position along the filtering axis is
    // a candidate as well
    int q2, c2;
    if ((VERTICAL || y - 4 >= d.y_begin) &&
        db_candidate(d, VERTICAL ? x - 4 : x, VERTICAL ? y : y - 4, VERTICAL, q2, c2))
      return;
  }
  for (;;) {
    db_filter_edge<VERTICAL>(d, pic, x, y, bs, qp, cqp);
    if (d.sub != 4) break;
    if (VERTICAL) x += 4; else y += 4;
    if (y >= d.y_end) break;
    bs = db_candidate(d, x, y, VERTICAL, qp, cqp);
    if (!bs) break;
  }
}

#endif  // XVCGPU_K_DEBLOCK_H_
