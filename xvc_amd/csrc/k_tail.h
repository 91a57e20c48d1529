// k_tail.h -- the tail of PictureEncoder::Encode (picture_encoder.cc:141-151)
// in ONE launch: DeblockingFilter::DeblockPicture (deblocking_filter.cc:56-450),
// YuvPicture::PadBorder (yuv_pic.cc:118-150) and the luma part of
// SampleMetric::ComparePicture (sample_metric.cc:37-155), for pictures whose
// CUs all lie on the 8-sample grid (every coding edge at a multiple of 8: no
// chains of interacting edges, k_deblock.h).
//
// The two-pass form (k_deblock.h) + pad + SSD is five latency-bound launches
// that each walk the picture.  Here the picture is cut into 64x64 tiles; a
// workgroup
//   1. loads its tile of the UNFILTERED reconstruction plus the halo the two
//      filter passes can see (luma: 4 rows above / below, 4 columns left /
//      right - fetched as 8 for 16-byte alignment; chroma likewise 2 -> 8) into
//      LDS, together with the CU records of the 10 x 10 cells of 8 x 8 samples
//      around it: two dependent round trips to memory in all - the candidate
//      test (CU map, CU records, boundary strength) of the 306 4-sample edge
//      segments the tile touches then runs from LDS (evaluated from HBM it is a
//      chain of five dependent loads per segment, which is what the two-pass
//      kernels spend their 8 us each on);
//   2. filters the vertical edges x = X0, X0+8, ..., X0+64 over rows
//      Y0-4 .. Y0+67 in LDS (the last edge and the halo rows belong to the
//      neighbours and are recomputed here: same inputs, same result);
//   3. filters the horizontal edges y = Y0 .. Y0+64 over its own columns;
//   4. writes its 64x64 samples (and its share of the border, if it touches the
//      picture's rim) to the OUTPUT picture and accumulates the squared error
//      against the original picture.
// Source and destination are different pictures: a tile reads unfiltered halo
// samples that a neighbour would otherwise already have replaced.
//
// ComparePicture's block walk is kept (picture_ssd_kernel, k_misc.h): a 64x64
// tile is one visited block, or - in the remainder column / row - a set of
// mbx x mby blocks, each shifted separately; the last block column / row of a
// dimension that is a multiple of 64 is not visited.  picture_ssd_sum_kernel
// folds the per-tile results.
#ifndef XVCGPU_K_TAIL_H_
#define XVCGPU_K_TAIL_H_

#include "k_deblock.h"

// diagnosis only (tools/micro/tail_time.hip builds variants that leave a step
// out, to see what the launch time is made of); 0 in the library
#ifndef TAIL_SKIP
#define TAIL_SKIP 0
#endif
#ifdef TAIL_TRACE   // the micro tool's per-workgroup clock readings (8 per tile)
__device__ unsigned long long *g_tail_trace;
#define TAIL_MARK(i)                                                            \
  do {                                                                          \
    if (threadIdx.x == 0) g_tail_trace[8 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define TAIL_MARK(i)
#endif

#define TAIL_LW 80  // luma LDS row:   columns X0-8 .. X0+71
#define TAIL_LH 72  //      LDS rows:  Y0-4 .. Y0+67
#define TAIL_CW 48  // chroma LDS row: columns CX0-8 .. CX0+39
#define TAIL_CH 36  //      LDS rows:  CY0-2 .. CY0+33

// The CU records of the 10 x 10 cells of 8 x 8 samples around the tile (cell
// (i, j) = samples X0-8+8i.., Y0-8+8j..; a CU of at least 8 x 8 covers whole
// cells): the boundary-strength inputs of every edge segment the tile filters.
struct alignas(16) TailShared {
  uint16_t luma[TAIL_LH * TAIL_LW];
  uint16_t chroma[2][TAIL_CH * TAIL_CW];
  xvcgpu_cu_info cell[100];
  int cell_idx[100];
  unsigned long long sub[64];
  uint8_t beta_tab[68], tc_tab[56];   // kBetaTable (+ the 0 of index 64), kTcTable
};

// db_candidate (k_deblock.h) on the tile's cell table instead of the CU map and
// the CU array in HBM: the candidate test of the 4-sample edge segment at (x, y)
// (kSubblockSizeExt, deblocking_filter.cc:59).
template <bool chroma>
__device__ __forceinline__ int tail_candidate(const TailShared &s, const DbParams &d, int X0,
                                              int Y0, int x, int y, bool vertical, int &qp,
                                              int &cqp) {
  if (x >= d.pic_w || y >= d.pic_h || x < 0 || y < 0) return 0;
  const int cq = ((y - Y0 + 8) >> 3) * 10 + ((x - X0 + 8) >> 3);
  const int cp = vertical ? cq - 1 : cq - 10;
  const int iq = s.cell_idx[cq], ip = s.cell_idx[cp];
  if (iq < 0 || ip < 0 || ip == iq) return 0;
  const xvcgpu_cu_info &p = s.cell[cp], &q = s.cell[cq];
  if (p.x == q.x && p.y == q.y) return 0;
  cqp = (p.qp_c + q.qp_c + 1) >> 1;
  if (chroma) return (p.intra || q.intra) ? 2 : 0;   // bs == 2 is all chroma asks
  const int bs = db_bs(d, p, q, x, y, vertical);
  qp = (p.qp_y + q.qp_y + 1) >> 1;
  return bs;
}

// FilterEdgeChroma for one line (deblocking_filter.cc:403-450): t points at
// q0, `off` = distance between samples across the edge.
__device__ __forceinline__ void tail_chroma_line(uint16_t *t, int off, int tc, int smax) {
  const int p1 = t[-2 * off], p0 = t[-off], q0 = t[0], q1 = t[off];
  const int delta = d_clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
  t[-off] = (uint16_t)d_clip_bd(p0 + delta, smax);
  t[0] = (uint16_t)d_clip_bd(q0 - delta, smax);
}


__device__ __forceinline__ int tail_chroma_tc(const TailShared &s, const DbParams &d, int cqp) {
  const int index_tc = d_clip3(cqp + d.tc_off + 2, 0, 54);
  return (int)s.tc_tab[index_tc] << (d.bd - 8);   // tc_tab[54] = 0
}

// The rim of a plane that a border tile owns: everything of the rectangle
// [ra, rb) x [ca, cb) (rows, 4-sample chunks, relative to the tile's own
// region) outside the own region [0, oh) x [0, ow/4), filled from the nearest
// own sample - PadBorder's result (rows copied, then columns replicated, so a
// corner takes the corner sample).
// CH = samples per store (8: 16-byte stores, when the own width is a multiple
// of 8; else 4).  A lane keeps its column of chunks; the lanes of a wave cover
// 64 / (row width rounded up to a power of two) rows per step, the four waves
// interleave.  What a lane stores above and below the picture does not depend on
// the row (the chunk of row 0 / row oh-1, or that row's edge sample in a border
// column): read once, then only stores; beside the picture a border column
// stores one edge sample per row.  (A row-major walk with an LDS read and index
// arithmetic per cell cost a corner tile 13 500 cycles - on the critical path of
// the launch - against 1 300 for the filters of a row of edges.)
template <int CH>
struct TailChunk;
template <>
struct TailChunk<8> { typedef uint4 T; };
template <>
struct TailChunk<4> { typedef uint2 T; };
__device__ __forceinline__ uint4 tail_splat(uint32_t e, uint4) {
  const uint32_t e2 = e | (e << 16);
  return make_uint4(e2, e2, e2, e2);
}
__device__ __forceinline__ uint2 tail_splat(uint32_t e, uint2) {
  const uint32_t e2 = e | (e << 16);
  return make_uint2(e2, e2);
}

template <int LW, int HX, int HY, int CH>
__device__ __forceinline__ void tail_pad_cells(const uint16_t *lds, const PlaneView dst, int x0,
                                               int y0, int ow, int oh, int tid, bool left,
                                               bool right, bool top, bool bottom) {
  typedef typename TailChunk<CH>::T V;
  const int B = dst.border, oc = ow / CH;
  const int ca = left ? -(B / CH) : 0, nc = oc - ca + (right ? B / CH : 0);
  const int lg = nc <= 8 ? 3 : (nc <= 16 ? 4 : (nc <= 32 ? 5 : (nc <= 64 ? 6 : 7)));
  const int cc = tid & ((1 << lg) - 1), ro = tid >> lg, step = 256 >> lg;
  const ptrdiff_t stride = dst.stride;
  for (int c0 = cc; c0 < nc; c0 += 1 << lg) {   // (one pass unless 64 < nc: 4-sample chunks)
    const int c = ca + c0;
    const bool own_col = c >= 0 && c < oc;
    const uint16_t *col = lds + HY * LW + HX + (c < 0 ? 0 : (c >= oc ? ow - 1 : CH * c));
    uint16_t *to = dst.p + (ptrdiff_t)y0 * stride + x0 + CH * c;
    if (top) {
      const V v = own_col ? *reinterpret_cast<const V *>(col) : tail_splat(col[0], V());
      uint16_t *out = to - (ptrdiff_t)(B - ro) * stride;
      for (int rr = ro; rr < B; rr += step, out += step * stride) *reinterpret_cast<V *>(out) = v;
    }
    if (bottom) {
      const uint16_t *last = col + (oh - 1) * LW;
      const V v = own_col ? *reinterpret_cast<const V *>(last) : tail_splat(last[0], V());
      uint16_t *out = to + (ptrdiff_t)(oh + ro) * stride;
      for (int rr = ro; rr < B; rr += step, out += step * stride) *reinterpret_cast<V *>(out) = v;
    }
    if (!own_col) {
      const uint16_t *from = col + ro * LW;
      uint16_t *out = to + (ptrdiff_t)ro * stride;
      int rr = ro;
      for (; rr + 3 * step < oh; rr += 4 * step, from += 4 * step * LW, out += 4 * step * stride) {
        const uint32_t e0 = from[0], e1 = from[step * LW], e2 = from[2 * step * LW],
                       e3 = from[3 * step * LW];
        *reinterpret_cast<V *>(out) = tail_splat(e0, V());
        *reinterpret_cast<V *>(out + step * stride) = tail_splat(e1, V());
        *reinterpret_cast<V *>(out + 2 * step * stride) = tail_splat(e2, V());
        *reinterpret_cast<V *>(out + 3 * step * stride) = tail_splat(e3, V());
      }
      for (; rr < oh; rr += step, from += step * LW, out += step * stride)
        *reinterpret_cast<V *>(out) = tail_splat(from[0], V());
    }
  }
}

template <int LW, int HX, int HY>
__device__ __forceinline__ void tail_pad_plane(const uint16_t *lds, const PlaneView dst, int x0,
                                               int y0, int ow, int oh, int tid) {
  const bool left = x0 == 0, right = x0 + ow == dst.w;
  const bool top = y0 == 0, bottom = y0 + oh == dst.h;
  if ((TAIL_SKIP & 8) || !(left || right || top || bottom)) return;
  if (ow & 7)
    tail_pad_cells<LW, HX, HY, 4>(lds, dst, x0, y0, ow, oh, tid, left, right, top, bottom);
  else
    tail_pad_cells<LW, HX, HY, 8>(lds, dst, x0, y0, ow, oh, tid, left, right, top, bottom);
}

// grid: tiles (ceil(w/64) * ceil(h/64)); block: 256.
// part: 2 words per tile (sum of the tile's block SSDs, samples visited), folded
// by picture_ssd_sum_kernel.  (Folding here - a ticket per workgroup, the last
// one sums - cost 10 us of the launch: 510 release fences + same-address
// returning atomics across the eight L2s; the second launch costs 3.)
template <bool SSD>
__device__ __forceinline__ void deblock_tail_kernel_body(DbParams d, PicView src, PicView dst, PlaneView orig, int shift, unsigned long long *part) {
  __shared__ TailShared s;
  const int tid = threadIdx.x;
  const int w = d.pic_w, h = d.pic_h;
  const int ntx = (w + 63) >> 6;
  // Tiles in raster order, XCD k takes the k-th contiguous eighth of them (the grid is
  // padded to a multiple of 8): a tile's halo - 4 rows / 8 columns of its neighbours'
  // samples, and the 128-byte lines they drag in: a row of 80 samples starting 16
  // bytes in front of a line boundary touches three lines for 160 bytes - then sits
  // in the L2 that serves those neighbours too.  With tile b on XCD b % 8 every
  // neighbour is on another XCD and each of the eight L2s fetches the shared lines
  // from memory for itself: 1.8x the algorithmic bytes (profiles/r03_traffic*.json).
  const int tile = xcd_job_index(blockIdx.x, ntx * ((h + 63) >> 6));
  if (tile < 0) return;
  const int tx = tile % ntx, ty = tile / ntx;
  const int X0 = tx << 6, Y0 = ty << 6, CX0 = X0 >> 1, CY0 = Y0 >> 1;
  const int ow = min(64, w - X0), oh = min(64, h - Y0);
  TAIL_MARK(0);

  // ---- 1. everything this workgroup reads: two dependent round trips ----
  // 1a. straight-line, nothing predicated (a predicated load makes the compiler
  // wait for it on the spot): a byte of the beta / tc tables and the CU map entry
  // of the thread's cell (two threads per cell, threads >= 200 repeat cell 99);
  // then - before anything else is requested, so that nothing queues in front of
  // them - the 11 / 10 words of the cell's CU record
  const int cell = min(tid >> 1, 99);
  const uint8_t tab_byte = tid < 64 ? kBetaTable[tid] : kTcTable[min(tid - 64, 53)];
  int cell_cu;
  {
    const int cj = cell / 10, ci = cell - 10 * cj;
    const int x = X0 - 8 + 8 * ci, y = Y0 - 8 + 8 * cj;
    const bool inside = !(TAIL_SKIP & 1) && x >= 0 && y >= 0 && x < w && y < h;
    cell_cu = d.map[inside ? (y >> 2) * d.map_stride + (x >> 2) : 0];
    if (!inside) cell_cu = -1;
  }
  uint32_t rec[11];
  {
    const uint32_t *from =
        reinterpret_cast<const uint32_t *>(d.cus + max(cell_cu, 0)) + ((tid & 1) ? 11 : 0);
#pragma unroll
    for (int k = 0; k < 11; k++) rec[k] = from[min(k, (tid & 1) ? 9 : 10)];
  }
  uint4 v[3], cv[2], ov[2];
  {
    const PlaneView pl = src.c[0];
    const uint16_t *base = pl.p + (ptrdiff_t)(Y0 - 4) * pl.stride + X0 - 8;
#pragma unroll
    for (int u = 0; u < 3; u++) {   // (index clamped, not predicated: every thread loads)
      const int i = min(tid + 256 * u, TAIL_LH * 10 - 1);
      const int r = i / 10, c = i - 10 * r;
      v[u] = make_uint4(0, 0, 0, 0);
      if (!(TAIL_SKIP & 64))
        v[u] = *reinterpret_cast<const uint4 *>(base + (ptrdiff_t)r * pl.stride + 8 * c);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int i = min(tid + 256 * u, 2 * TAIL_CH * 6 - 1);   // 2 planes x 36 rows x 6 chunks
      const int p = i >= TAIL_CH * 6, j = i - p * TAIL_CH * 6;
      const int r = j / 6, c = j - 6 * r;
      const uint16_t *pp = p ? src.c[2].p : src.c[1].p;
      cv[u] = make_uint4(0, 0, 0, 0);
      if (!(TAIL_SKIP & 64))
        cv[u] = *reinterpret_cast<const uint4 *>(
            pp + (ptrdiff_t)(CY0 - 2 + r) * src.c[1].stride + CX0 - 8 + 8 * c);
    }
  }
  if (SSD) {   // the original samples this thread compares in step 4 (rows r, r + 32);
               // row / column clamped, not predicated
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int r = min((tid >> 3) + 32 * u, oh - 1), c8 = min((tid & 7) << 3, ow - 8);
      ov[u] = *reinterpret_cast<const uint4 *>(orig.p + (ptrdiff_t)(Y0 + r) * orig.stride +
                                               X0 + c8);
    }
  }
  if (tid < 64) s.sub[tid] = 0;
  if (tid < 200) {   // two threads per cell: 11 + 10 words of the record
    if (!(tid & 1)) s.cell_idx[cell] = cell_cu;
    uint32_t *to = reinterpret_cast<uint32_t *>(&s.cell[cell]) + ((tid & 1) ? 11 : 0);
#pragma unroll
    for (int k = 0; k < 11; k++)
      if (k < ((tid & 1) ? 10 : 11)) to[k] = rec[k];
  }
  if (tid < 64) s.beta_tab[tid] = tab_byte;
  else if (tid < 64 + 54) s.tc_tab[tid - 64] = tab_byte;
  else if (tid < 64 + 56) s.tc_tab[tid - 64] = 0;       // index 54: tc = 0
  else if (tid < 64 + 60) s.beta_tab[tid - 56] = 0;     // index 64: beta = 0
#pragma unroll
  for (int u = 0; u < 3; u++) {
    const int i = tid + 256 * u;
    if (i < TAIL_LH * 10) reinterpret_cast<uint4 *>(s.luma)[i] = v[u];
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int i = tid + 256 * u;
    if (i < 2 * TAIL_CH * 6) reinterpret_cast<uint4 *>(&s.chroma[0][0])[i] = cv[u];
  }
  __syncthreads();
  TAIL_MARK(1);

  // ---- 2. vertical edges ----
  if (!(TAIL_SKIP & 2) && tid < 162) {
    const int k = tid / 18, g = tid - 18 * k;
    int qp = 0, cqp = 0;
    const int bs = tail_candidate<false>(s, d, X0, Y0, X0 + 8 * k, Y0 - 4 + 4 * g, true, qp, cqp);
    if (bs) {
      uint16_t *base = s.luma + (4 * g) * TAIL_LW + 8 * k + 4;
      int t[4][8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint2 *r = reinterpret_cast<const uint2 *>(base + i * TAIL_LW);
        db_unpack8(r[0], r[1], t[i]);
      }
      const int beta = (int)s.beta_tab[db_beta_index(qp, d.beta_off)] << (d.bd - 8);
      const int tc = (int)s.tc_tab[db_tc_index(qp, d.tc_off, bs)] << (d.bd - 8);
      if (db_filter_luma_group_bt(t, d.bd, beta, tc)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          uint2 *r = reinterpret_cast<uint2 *>(base + i * TAIL_LW);
          r[0] = make_uint2(db_pack(t[i][0], t[i][1]), db_pack(t[i][2], t[i][3]));
          r[1] = make_uint2(db_pack(t[i][4], t[i][5]), db_pack(t[i][6], t[i][7]));
        }
      }
    }
  }
  const int smax = (1 << d.bd) - 1;
  for (int i = tid; !(TAIL_SKIP & 2) && i < 2 * 5 * TAIL_CH; i += 256) {
    const int p = i >= 5 * TAIL_CH, j = i - p * 5 * TAIL_CH;
    const int kc = j / TAIL_CH, r = j - kc * TAIL_CH;
    int qp = 0, cqp = 0;
    if (tail_candidate<true>(s, d, X0, Y0, X0 + 16 * kc, Y0 - 4 + 4 * (r >> 1), true, qp, cqp) == 2)
      tail_chroma_line(&s.chroma[p][r * TAIL_CW + 8 * kc + 8], 1, tail_chroma_tc(s, d, cqp), smax);
  }
  __syncthreads();
  TAIL_MARK(2);

  // ---- 3. horizontal edges ----
  if (!(TAIL_SKIP & 4) && tid < 144) {
    const int k = tid >> 4, g = tid & 15;
    int qp = 0, cqp = 0;
    const int bs = tail_candidate<false>(s, d, X0, Y0, X0 + 4 * g, Y0 + 8 * k, false, qp, cqp);
    if (bs) {
      uint16_t *base = s.luma + (8 * k) * TAIL_LW + 8 + 4 * g;
      int t[4][8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const uint2 q = *reinterpret_cast<const uint2 *>(base + r * TAIL_LW);
        t[0][r] = q.x & 0xffff;
        t[1][r] = q.x >> 16;
        t[2][r] = q.y & 0xffff;
        t[3][r] = q.y >> 16;
      }
      const int beta = (int)s.beta_tab[db_beta_index(qp, d.beta_off)] << (d.bd - 8);
      const int tc = (int)s.tc_tab[db_tc_index(qp, d.tc_off, bs)] << (d.bd - 8);
      if (db_filter_luma_group_bt(t, d.bd, beta, tc)) {
#pragma unroll
        for (int r = 1; r < 7; r++)
          *reinterpret_cast<uint2 *>(base + r * TAIL_LW) =
              make_uint2(db_pack(t[0][r], t[1][r]), db_pack(t[2][r], t[3][r]));
      }
    }
  }
  for (int i = tid; !(TAIL_SKIP & 4) && i < 2 * 5 * 32; i += 256) {
    const int p = i >= 160, j = i - 160 * p;
    const int kc = j >> 5, cc = j & 31;
    int qp = 0, cqp = 0;
    if (tail_candidate<true>(s, d, X0, Y0, X0 + 4 * (cc >> 1), Y0 + 16 * kc, false, qp, cqp) == 2)
      tail_chroma_line(&s.chroma[p][(8 * kc + 2) * TAIL_CW + 8 + cc], TAIL_CW,
                       tail_chroma_tc(s, d, cqp), smax);
  }
  __syncthreads();
  TAIL_MARK(3);

  // ---- 4. own samples out, squared error, rim ----
  // ComparePicture's walk over this tile (see picture_ssd_kernel)
  const int mbx = w & ~(w - 1), mby = h & ~(h - 1);
  const int nfx = w > 64 ? (w - 1) >> 6 : 0, nfy = h > 64 ? (h - 1) >> 6 : 0;
  const bool full_x = tx < nfx, full_y = ty < nfy;
  const bool vis = (full_x || (X0 == (w & ~63) && (w & 63))) &&
                   (full_y || (Y0 == (h & ~63) && (h & 63)));
  const int lbx = 31 - __clz(mbx), lby = 31 - __clz(mby);
  // squared error first, while no store is in flight: the loads of the original
  // are consumed on every path here (a load the compiler cannot prove finished
  // makes it wait for ALL memory operations - the stores too, they share the
  // counter on gfx9 - wherever its registers are reused: that cost the rim
  // tiles 5 us)
  uint4 lv[2];
  uint32_t acc[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int r = min((tid >> 3) + 32 * u, oh - 1), c8 = min((tid & 7) << 3, ow - 8);
    lv[u] = *reinterpret_cast<const uint4 *>(s.luma + (r + 4) * TAIL_LW + 8 + c8);
    acc[u] = 0;
    if (SSD) {
      const uint32_t ua[4] = {lv[u].x, lv[u].y, lv[u].z, lv[u].w};
      const uint32_t ub[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int d0 = (int)(ua[k] & 0xffff) - (int)(ub[k] & 0xffff);
        const int d1 = (int)(ua[k] >> 16) - (int)(ub[k] >> 16);
        acc[u] += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
      }
    }
  }
  unsigned long long acc_full = 0;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int r = (tid >> 3) + 32 * u, c8 = (tid & 7) << 3;
    if (r < oh && c8 < ow) {
      if (!(TAIL_SKIP & 32))
        *reinterpret_cast<uint4 *>(dst.c[0].p + (ptrdiff_t)(Y0 + r) * dst.c[0].stride + X0 + c8) =
            lv[u];
      if (SSD && vis) {
        if (full_x && full_y) {
          acc_full += acc[u];
        } else {
          const int bx = full_x ? 0 : c8 >> lbx, by = full_y ? 0 : r >> lby;
          atomicAdd(&s.sub[(by << 3) + bx], (unsigned long long)acc[u]);
        }
      }
    }
  }
  if (SSD && vis && full_x && full_y) {   // uniform
    acc_full = group_sum<64>(acc_full);
    if ((tid & 63) == 0) atomicAdd(&s.sub[0], acc_full);
  }
#pragma unroll
  for (int p = 0; p < 2; p++) {           // chroma: 32 rows x 8 chunks of 4
    const int r = tid >> 3, c4 = (tid & 7) << 2;
    const PlaneView pc = dst.c[1 + p];
    if (!(TAIL_SKIP & 32) && r < (oh >> 1) && c4 < (ow >> 1))
      *reinterpret_cast<uint2 *>(pc.p + (ptrdiff_t)(CY0 + r) * pc.stride + CX0 + c4) =
          *reinterpret_cast<const uint2 *>(&s.chroma[p][(r + 2) * TAIL_CW + 8 + c4]);
  }
  TAIL_MARK(4);
  tail_pad_plane<TAIL_LW, 8, 4>(s.luma, dst.c[0], X0, Y0, ow, oh, tid);
  tail_pad_plane<TAIL_CW, 8, 2>(s.chroma[0], dst.c[1], CX0, CY0, ow >> 1, oh >> 1, tid);
  tail_pad_plane<TAIL_CW, 8, 2>(s.chroma[1], dst.c[2], CX0, CY0, ow >> 1, oh >> 1, tid);
  TAIL_MARK(5);
  if (!SSD) return;

  __syncthreads();
  if (tid == 0) {
    unsigned long long sum = 0;
    if (vis) {
      const int nbx = full_x ? 1 : ow >> lbx, nby = full_y ? 1 : oh >> lby;
      for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) sum += s.sub[(by << 3) + bx] >> shift;
    }
    part[2 * tile] = sum;
    part[2 * tile + 1] = vis ? (unsigned long long)ow * oh : 0ull;
  }
  TAIL_MARK(6);
}

// Waves per SIMD the tail is compiled for (registers: 136 at 3, 128 at 4 with 8 spilled
// dwords, 96 at 5 with 44).  Four pictures per launch, pictures not in the Infinity Cache
// (tools/tail_batched.sh), share of 8 TB/s at 1080p / 4320p: 3: 26.0 / 33.2 %, 4: 29.5 /
// 39.7 %, 5: 21.4 / 25.8 %; 4320p frame pass 697 -> 703 passes/s at 4.
#ifndef TAIL_MIN_WAVES
#define TAIL_MIN_WAVES 4
#endif
template <bool SSD>
__global__ void __launch_bounds__(256, TAIL_MIN_WAVES)
deblock_tail_kernel(DbParams d, PicView src, PicView dst, PlaneView orig, int shift, unsigned long long *part) {
  deblock_tail_kernel_body<SSD>(d, src, dst, orig, shift, part);
}

#endif  // XVCGPU_K_TAIL_H_
