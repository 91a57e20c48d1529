// k_me.h -- T1 + T3 (+W1, I1, M1, M4): one workgroup = one call of
// InterSearch::MotionEstNormal with the TZ search method
// (inter_search.cc:606-662): TzSearch::Search (inter_tz_search.cc:84-171)
// followed by SubpelSearch (inter_search.cc:893-964).
//
// Exactness argument: every selection in
// the reference is a left-to-right fold `if (cost < best) best = cost` with
// cost = dist + ((lambda16*bits) >> 16) >= dist, so the early
// `dist >= best -> skip` never changes the outcome and the candidate costs of
// one phase do not depend on the running state.  Each phase therefore
// (1) materialises its candidate list, (2) evaluates all costs in parallel
// (16 groups of 16 lanes, one candidate per group at a time) and
// (3) replays the reference's sequential control flow over the cost array,
// uniformly in every thread - including the per-round `changed` /
// `rounds_with_no_match` logic, `last_position`, `last_range_` and the single
// bound test of CheckCost1/2.  Out-of-window candidates (zero MV, previous
// CU's MV) are read from global memory like any other: there is no LDS window.
//
// Data movement: the original block is staged once in LDS (16-byte rows); the
// reference picture is read through L1/L2 with 16-byte unaligned loads of 8
// samples per lane, v_sad_u16 on packed pairs, DPP/xor-shuffle group sums.
#ifndef XVCGPU_K_ME_H_
#define XVCGPU_K_ME_H_

#include "dev_common.h"
#include "k_interp.h"
#include "k_metric.h"
#include "xvcgpu_internal.h"

#define ME_THREADS 256
#define ME_GROUPS 16  // 16 lanes each
#define ME_MAX_LIST 128

struct __attribute__((packed, aligned(2))) U16x8 {
  uint32_t v[4];
};
struct __attribute__((packed, aligned(2))) U16x4 {
  uint32_t v[2];
};

struct MeShared {
  uint16_t orig[64 * 64];   // original block, row stride = w
  uint16_t pred[64 * 64];   // interpolated candidate, row stride = w
  int16_t tmp[64 * 71];     // 14-bit intermediate of the separable filter
  int16_t cx[ME_MAX_LIST], cy[ME_MAX_LIST];
  int8_t cpos[ME_MAX_LIST], cvalid[ME_MAX_LIST];
  int16_t crng[ME_MAX_LIST];
  uint32_t ccost[ME_MAX_LIST];
  uint32_t gcost[ME_GROUPS];
  int32_t gidx[ME_GROUPS];
};

struct MeCtx {
  int bd, w, h, rows, row_step, sad_shift, sad_mul;  // kSad / kSadFast
  const uint16_t *ref;  // reference plane at the CU position
  int rs;
  int mvp_x, mvp_y, down;
  uint32_t lambda;
  int min_x, min_y, max_x, max_y;  // TZ window (mv_min / mv_max)
};

// SAD of the LDS-resident original block against the reference block at
// full-pel displacement (mx,my), computed by one 16-lane group.  Returns the
// metric value after the kSad / kSadFast post-scaling (sample_metric.cc:
// 189-199) in every lane of the group.
__device__ __forceinline__ uint32_t group_sad(const MeCtx &c,
                                              const uint16_t *s_orig, int mx,
                                              int my, int sub) {
  const uint16_t *r = c.ref + (ptrdiff_t)my * c.rs + mx;
  uint32_t sum = 0;
  if (c.w >= 8) {
    const int spr = c.w >> 3;  // 8-sample segments per row
    const int nseg = c.rows * spr;
    const int lspr = 31 - __clz(spr);
    for (int i = sub; i < nseg; i += 16) {
      const int y = (i >> lspr) * c.row_step, x = (i & (spr - 1)) << 3;
      const uint4 a = *reinterpret_cast<const uint4 *>(s_orig + y * c.w + x);
      const U16x8 b = *reinterpret_cast<const U16x8 *>(r + (ptrdiff_t)y * c.rs + x);
      sum = __builtin_amdgcn_sad_u16(a.x, b.v[0], sum);
      sum = __builtin_amdgcn_sad_u16(a.y, b.v[1], sum);
      sum = __builtin_amdgcn_sad_u16(a.z, b.v[2], sum);
      sum = __builtin_amdgcn_sad_u16(a.w, b.v[3], sum);
    }
  } else {  // w == 4
    for (int y = sub; y < c.rows; y += 16) {
      const int yy = y * c.row_step;
      const uint2 a = *reinterpret_cast<const uint2 *>(s_orig + yy * 4);
      const U16x4 b = *reinterpret_cast<const U16x4 *>(r + (ptrdiff_t)yy * c.rs);
      sum = __builtin_amdgcn_sad_u16(a.x, b.v[0], sum);
      sum = __builtin_amdgcn_sad_u16(a.y, b.v[1], sum);
    }
  }
  sum = group_sum<16>(sum);
  return (sum * c.sad_mul) >> c.sad_shift;
}

__device__ __forceinline__ uint32_t me_cost(const MeCtx &c, uint32_t dist,
                                            int mx, int my) {
  const uint32_t bits = d_mvd_bits_fullpel(c.mvp_x, c.mvp_y, mx, my, c.down);
  return dist + ((c.lambda * bits) >> 16);
}

// Evaluate ccost[i] for the n listed candidates (cvalid[i] != 0).
__device__ __forceinline__ void me_eval_list(const MeCtx &c, MeShared &s, int n) {
  __syncthreads();  // list written
  const int g = threadIdx.x >> 4, sub = threadIdx.x & 15;
  for (int i = g; i < n; i += ME_GROUPS) {
    if (!s.cvalid[i]) continue;
    const int mx = s.cx[i], my = s.cy[i];
    const uint32_t dist = group_sad(c, s.orig, mx, my, sub);
    if (sub == 0) s.ccost[i] = me_cost(c, dist, mx, my);
  }
  __syncthreads();  // costs visible
}

enum { TZ_LEFT = -1, TZ_RIGHT = 1, TZ_UP = -3, TZ_DOWN = 3 };

__device__ __forceinline__ bool tz_inside(const MeCtx &c, int dir, int mx, int my) {
  switch (dir) {
    case TZ_UP: return my >= c.min_y;
    case TZ_DOWN: return my <= c.max_y;
    case TZ_LEFT: return mx >= c.min_x;
    default: return mx <= c.max_x;
  }
}

// k-th candidate of FullpelDiamondSearch(base, range) in issue order
// (inter_tz_search.cc:173-210).
__device__ __forceinline__ void tz_diamond_cand(const MeCtx &c, int bx, int by,
                                                int range, int k, int &x,
                                                int &y, int &pos, int &rng,
                                                bool &valid) {
  int d1 = 0, d2 = 0;
  if (range == 1) {
    const int dirs[4] = {TZ_UP, TZ_LEFT, TZ_RIGHT, TZ_DOWN};
    d1 = dirs[k];
    rng = range;
    x = bx + (d1 == TZ_LEFT ? -range : d1 == TZ_RIGHT ? range : 0);
    y = by + (d1 == TZ_UP ? -range : d1 == TZ_DOWN ? range : 0);
  } else if (range <= 8) {
    const int r2 = range >> 1;
    switch (k) {
      case 0: d1 = TZ_UP; x = bx; y = by - range; rng = range; break;
      case 1: d1 = TZ_UP; d2 = TZ_LEFT; x = bx - r2; y = by - r2; rng = r2; break;
      case 2: d1 = TZ_UP; d2 = TZ_RIGHT; x = bx + r2; y = by - r2; rng = r2; break;
      case 3: d1 = TZ_LEFT; x = bx - range; y = by; rng = range; break;
      case 4: d1 = TZ_RIGHT; x = bx + range; y = by; rng = range; break;
      case 5: d1 = TZ_DOWN; d2 = TZ_LEFT; x = bx - r2; y = by + r2; rng = r2; break;
      case 6: d1 = TZ_DOWN; d2 = TZ_RIGHT; x = bx + r2; y = by + r2; rng = r2; break;
      default: d1 = TZ_DOWN; x = bx; y = by + range; rng = range; break;
    }
  } else {
    rng = range;
    if (k < 4) {
      const int dirs[4] = {TZ_UP, TZ_LEFT, TZ_RIGHT, TZ_DOWN};
      d1 = dirs[k];
      x = bx + (d1 == TZ_LEFT ? -range : d1 == TZ_RIGHT ? range : 0);
      y = by + (d1 == TZ_UP ? -range : d1 == TZ_DOWN ? range : 0);
    } else {
      const int i = 1 + ((k - 4) >> 2), q = (k - 4) & 3;
      const int r14 = i * (range >> 2), r34 = range - r14;
      d1 = (q < 2) ? TZ_UP : TZ_DOWN;
      d2 = (q & 1) ? TZ_RIGHT : TZ_LEFT;
      x = bx + ((q & 1) ? r14 : -r14);
      y = by + ((q < 2) ? -r34 : r34);
    }
  }
  pos = d1 + d2;
  valid = tz_inside(c, d1, x, y) && (d2 == 0 || tz_inside(c, d2, x, y));
}

__device__ __forceinline__ int tz_diamond_count(int range) {
  return range == 1 ? 4 : (range <= 8 ? 8 : 16);
}

struct TzState {
  int bx, by;
  uint32_t cost;
  int last_pos, last_range;
};

// Build the list of all diamonds around (bx,by) for ranges 1,2,4,..<=R.
// Returns the number of candidates.
__device__ __forceinline__ int tz_build_diamonds(const MeCtx &c, MeShared &s,
                                                 int bx, int by, int R) {
  int total = 0;
  for (int range = 1; range <= R; range *= 2) total += tz_diamond_count(range);
  const int i = threadIdx.x;
  if (i < total) {
    int off = 0, range = 1;
    while (i >= off + tz_diamond_count(range)) {
      off += tz_diamond_count(range);
      range *= 2;
    }
    int x, y, pos, rng;
    bool valid;
    tz_diamond_cand(c, bx, by, range, i - off, x, y, pos, rng, valid);
    s.cx[i] = (int16_t)x;
    s.cy[i] = (int16_t)y;
    s.cpos[i] = (int8_t)pos;
    s.crng[i] = (int16_t)rng;
    s.cvalid[i] = valid;
  }
  return total;
}

// Replay of CheckCost1/2 over list entry i (inter_tz_search.cc:304-336).
__device__ __forceinline__ bool tz_take(const MeShared &s, TzState &st, int i) {
  if (!s.cvalid[i]) return false;
  const uint32_t cost = s.ccost[i];
  if (cost < st.cost) {
    st.cost = cost;
    st.bx = s.cx[i];
    st.by = s.cy[i];
    st.last_pos = s.cpos[i];
    st.last_range = s.crng[i];
    return true;
  }
  return false;
}

// FullpelNeighborPointSearch (inter_tz_search.cc:212-259): two candidates.
__device__ __forceinline__ void tz_neighbor(const MeCtx &c, MeShared &s,
                                            TzState &st) {
  const int r = 1, bx = st.bx, by = st.by;
  int x[2], y[2], d1[2], d2[2];
  int n = 2;
  switch (st.last_pos) {
    case TZ_UP + TZ_LEFT:
      x[0] = bx - r; y[0] = by; d1[0] = TZ_LEFT; d2[0] = 0;
      x[1] = bx; y[1] = by - r; d1[1] = TZ_UP; d2[1] = 0;
      break;
    case TZ_UP:
      x[0] = bx - r; y[0] = by - r; d1[0] = TZ_UP; d2[0] = TZ_LEFT;
      x[1] = bx + r; y[1] = by - r; d1[1] = TZ_UP; d2[1] = TZ_RIGHT;
      break;
    case TZ_UP + TZ_RIGHT:
      x[0] = bx; y[0] = by - r; d1[0] = TZ_UP; d2[0] = 0;
      x[1] = bx + r; y[1] = by; d1[1] = TZ_RIGHT; d2[1] = 0;
      break;
    case TZ_LEFT:
      x[0] = bx - r; y[0] = by + r; d1[0] = TZ_DOWN; d2[0] = TZ_LEFT;
      x[1] = bx - r; y[1] = by - r; d1[1] = TZ_UP; d2[1] = TZ_LEFT;
      break;
    case TZ_RIGHT:
      x[0] = bx + r; y[0] = by - r; d1[0] = TZ_UP; d2[0] = TZ_RIGHT;
      x[1] = bx + r; y[1] = by + r; d1[1] = TZ_DOWN; d2[1] = TZ_RIGHT;
      break;
    case TZ_DOWN + TZ_LEFT:
      x[0] = bx - r; y[0] = by; d1[0] = TZ_LEFT; d2[0] = 0;
      x[1] = bx; y[1] = by + r; d1[1] = TZ_DOWN; d2[1] = 0;
      break;
    case TZ_DOWN:
      x[0] = bx - r; y[0] = by + r; d1[0] = TZ_DOWN; d2[0] = TZ_LEFT;
      x[1] = bx + r; y[1] = by + r; d1[1] = TZ_DOWN; d2[1] = TZ_RIGHT;
      break;
    case TZ_DOWN + TZ_RIGHT:
      x[0] = bx + r; y[0] = by; d1[0] = TZ_RIGHT; d2[0] = 0;
      x[1] = bx; y[1] = by + r; d1[1] = TZ_DOWN; d2[1] = 0;
      break;
    default:
      n = 0;
      break;
  }
  if (n == 0) return;
  __syncthreads();  // previous fold finished reading the list
  if (threadIdx.x < 2) {
    const int i = threadIdx.x;
    s.cx[i] = (int16_t)x[i];
    s.cy[i] = (int16_t)y[i];
    s.cpos[i] = (int8_t)(d1[i] + d2[i]);
    s.crng[i] = (int16_t)r;
    s.cvalid[i] = tz_inside(c, d1[i], x[i], y[i]) &&
                  (d2[i] == 0 || tz_inside(c, d2[i], x[i], y[i]));
  }
  me_eval_list(c, s, 2);
  tz_take(s, st, 0);
  tz_take(s, st, 1);
}

// CheckCostBest on a single MV (initial predictor / zero / previous CU).
__device__ __forceinline__ bool tz_check_single(const MeCtx &c, MeShared &s,
                                                TzState &st, int mx, int my) {
  __syncthreads();
  if (threadIdx.x == 0) {
    s.cx[0] = (int16_t)mx;
    s.cy[0] = (int16_t)my;
    s.cvalid[0] = 1;
  }
  me_eval_list(c, s, 1);
  const uint32_t cost = s.ccost[0];
  if (cost < st.cost) {
    st.cost = cost;
    st.bx = mx;
    st.by = my;
    return true;
  }
  return false;
}

// DetermineMinMaxMv (inter_prediction.cc:801-817), full-pel result.
__device__ __forceinline__ void d_min_max_mv(int px, int py, int pw, int ph,
                                             int cx, int cy, int range,
                                             int &mnx, int &mny, int &mxx,
                                             int &mxy) {
  d_clip_mv(px, py, pw, ph, cx, cy);
  const int r = range << 4;
  int a = cx - r, b = cy - r, e = cx + r, f = cy + r;
  d_clip_mv(px, py, pw, ph, a, b);
  d_clip_mv(px, py, pw, ph, e, f);
  mnx = a >> 4;
  mny = b >> 4;
  mxx = e >> 4;
  mxy = f >> 4;
}

// ---- sub-pel refinement ----------------------------------------------------
// One candidate: MotionCompensationMv (clip, split, interpolate into LDS) and
// SATD against the original block, by the whole workgroup.
__device__ __forceinline__ uint32_t me_subpel_dist(const MeCtx &c, MeShared &s,
                                                   const xvcgpu_me_block &b,
                                                   int pic_w, int pic_h,
                                                   const uint16_t *ref_plane,
                                                   int mx, int my) {
  d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);
  const uint16_t *r =
      ref_plane + (ptrdiff_t)(b.y + (my >> 4)) * c.rs + b.x + (mx >> 4);
  __syncthreads();  // previous candidate's SATD has finished reading pred/tmp
  wg_interp_block<false>(c.bd, c.w, c.h, mx & 15, my & 15, r, c.rs, s.tmp,
                         s.pred, c.w);
  __syncthreads();
  // SATD: tiles spread over the 4 waves; each wave handles whole rows of
  // tiles so wave_satd() can be reused on a horizontal strip.
  const int wave = threadIdx.x >> 6;
  // strip height: tile height of the (w,h) SATD tiling
  int th;
  if (c.w == 4 && c.h == 4) th = 4;
  else if (c.h == 4 && c.w > c.h) th = 4;
  else if (c.w == 4 && c.h > c.w) th = 8;
  else if (c.w > c.h) th = 8;
  else if (c.w < c.h) th = 16;
  else th = 8;
  // The tile shape depends on the (w,h) relation of the WHOLE block, so the
  // strip is evaluated with the block-level tile choice via explicit dispatch.
  const int n_strips = c.h / th;
  uint32_t part = 0;
  for (int st = wave; st < n_strips; st += 4) {
    const uint16_t *a = s.orig + st * th * c.w;
    const uint16_t *p = s.pred + st * th * c.w;
    uint64_t v;
    if (c.w == 4 && c.h == 4) v = wave_satd_tiles<4, 4>(c.w, th, 0, a, c.w, p, c.w);
    else if (c.h == 4 && c.w > c.h) v = wave_satd_tiles<8, 4>(c.w, th, 0, a, c.w, p, c.w);
    else if (c.w == 4 && c.h > c.w) v = wave_satd_tiles<4, 8>(c.w, th, 0, a, c.w, p, c.w);
    else if (c.w > c.h) v = wave_satd_tiles<16, 8>(c.w, th, 0, a, c.w, p, c.w);
    else if (c.w < c.h) v = wave_satd_tiles<8, 16>(c.w, th, 0, a, c.w, p, c.w);
    else v = wave_satd_tiles<8, 8>(c.w, th, 0, a, c.w, p, c.w);
    part += (uint32_t)v;
  }
  if ((threadIdx.x & 63) == 0) s.gcost[wave] = part;
  __syncthreads();
  const uint32_t total = s.gcost[0] + s.gcost[1] + s.gcost[2] + s.gcost[3];
  return total >> (c.bd - 8);
}

// grid: n blocks; block: 256 threads.
__global__ void __launch_bounds__(ME_THREADS)
me_search_kernel(PicView orig, PicView ref, int flags,
                 const xvcgpu_me_block *blocks, int n,
                 xvcgpu_me_result *results) {
  __shared__ MeShared s;
  const int bi = blockIdx.x;
  if (bi >= n) return;
  const xvcgpu_me_block b = blocks[bi];
  const PlaneView po = orig.c[0], pr = ref.c[0];
  const int pic_w = po.w, pic_h = po.h;

  MeCtx c;
  c.bd = orig.bd;
  c.w = b.w;
  c.h = b.h;
  const bool fast = b.h > 8;  // GetFullpelMetric, inter_search.cc:1059-1069
  c.rows = fast ? b.h / 2 : b.h;
  c.row_step = fast ? 2 : 1;
  c.sad_mul = fast ? 2 : 1;
  c.sad_shift = c.bd - 8;
  c.rs = pr.stride;
  c.ref = pr.p + (ptrdiff_t)b.y * pr.stride + b.x;
  c.mvp_x = b.mvp_x;
  c.mvp_y = b.mvp_y;
  c.down = b.fullpel_mv ? 2 : 0;
  c.lambda = b.lambda16;

  // stage the original block (row stride w) in LDS
  {
    const int lw = 31 - __clz(c.w);
    const uint16_t *o = po.p + (ptrdiff_t)b.y * po.stride + b.x;
    for (int i = threadIdx.x; i < c.w * c.h; i += ME_THREADS) {
      const int y = i >> lw, x = i & (c.w - 1);
      s.orig[i] = o[(ptrdiff_t)y * po.stride + x];
    }
  }

  xvcgpu_me_result res;
  if (flags & XVCGPU_ME_FULLPEL) {
    const int range = b.search_range;
    d_min_max_mv(b.x, b.y, pic_w, pic_h, b.mvp_x, b.mvp_y, range, c.min_x,
                 c.min_y, c.max_x, c.max_y);
    int fs_min_x = c.min_x, fs_min_y = c.min_y, fs_max_x = c.max_x,
        fs_max_y = c.max_y;
    TzState st;
    st.bx = 0;
    st.by = 0;
    st.cost = 0xffffffffu;
    st.last_pos = 0;
    st.last_range = 0;

    int mx = b.mvp_x, my = b.mvp_y;
    d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);
    tz_check_single(c, s, st, mx >> 4, my >> 4);
    bool change_min_max = false;
    if (st.bx != 0 || st.by != 0) change_min_max = tz_check_single(c, s, st, 0, 0);
    st.last_range = 0;
    if (b.depth_nonzero) {
      int px = b.prev_x * 16, py = b.prev_y * 16;
      d_clip_mv(b.x, b.y, pic_w, pic_h, px, py);
      change_min_max |= tz_check_single(c, s, st, px >> 4, py >> 4);
      if (change_min_max)
        d_min_max_mv(b.x, b.y, pic_w, pic_h, st.bx * 16, st.by * 16, range,
                     fs_min_x, fs_min_y, fs_max_x, fs_max_y);
    }

    // initial raster around the fixed base: all ranges evaluated at once, then
    // the reference's per-round early-termination logic replayed.
    {
      __syncthreads();
      const int total = tz_build_diamonds(c, s, st.bx, st.by, range);
      me_eval_list(c, s, total);
      int off = 0, no_match = 0;
      for (int r = 1; r <= range; r *= 2) {
        const int cnt = tz_diamond_count(r);
        bool changed = false;
        for (int k = 0; k < cnt; k++) changed |= tz_take(s, st, off + k);
        off += cnt;
        if (changed) {
          no_match = 0;
        } else if (++no_match >= 3) {
          break;
        }
      }
    }
    if (st.last_range == 1) {
      st.last_range = 0;
      tz_neighbor(c, s, st);
    }
    // step-5 grid over the (possibly re-centred) window
    if (st.last_range > 5) {
      st.last_range = 5;
      const int nx = (fs_max_x - fs_min_x) / 5 + 1;
      const int ny = (fs_max_y - fs_min_y) / 5 + 1;
      const int total = (fs_max_x >= fs_min_x && fs_max_y >= fs_min_y) ? nx * ny : 0;
      const int g = threadIdx.x >> 4, sub = threadIdx.x & 15;
      uint32_t best = 0xffffffffu;
      int best_i = 0x7fffffff;
      for (int i = g; i < total; i += ME_GROUPS) {
        const int gx = fs_min_x + (i % nx) * 5, gy = fs_min_y + (i / nx) * 5;
        const uint32_t cost = me_cost(c, group_sad(c, s.orig, gx, gy, sub), gx, gy);
        if (cost < best) {
          best = cost;
          best_i = i;
        }
      }
      __syncthreads();
      if (sub == 0) {
        s.gcost[g] = best;
        s.gidx[g] = best_i;
      }
      __syncthreads();
      uint32_t gb = 0xffffffffu;
      int gi = 0x7fffffff;
      for (int k = 0; k < ME_GROUPS; k++) {
        const uint32_t cc = s.gcost[k];
        const int ii = s.gidx[k];
        if (cc < gb || (cc == gb && ii < gi)) {
          gb = cc;
          gi = ii;
        }
      }
      if (gb < st.cost) {
        st.cost = gb;
        st.bx = fs_min_x + (gi % nx) * 5;
        st.by = fs_min_y + (gi / nx) * 5;
      }
    }
    // iterative refinement: every round evaluates all diamonds around the
    // current best (no early termination in the reference here).
    while (st.last_range > 0) {
      st.last_range = 0;
      __syncthreads();
      const int total = tz_build_diamonds(c, s, st.bx, st.by, range);
      me_eval_list(c, s, total);
      for (int k = 0; k < total; k++) tz_take(s, st, k);
      if (st.last_range == 1) {
        st.last_range = 0;
        tz_neighbor(c, s, st);
      }
    }
    res.fullpel_x = st.bx;
    res.fullpel_y = st.by;
    res.fullpel_cost = st.cost;
  } else {
    res.fullpel_x = results[bi].fullpel_x;
    res.fullpel_y = results[bi].fullpel_y;
    res.fullpel_cost = results[bi].fullpel_cost;
  }
  res.mv_x = res.fullpel_x * 16;
  res.mv_y = res.fullpel_y * 16;
  res.subpel_dist = 0;

  if (flags & XVCGPU_ME_SUBPEL) {
    if (b.fullpel_mv) {
      // cu.GetFullpelMv(): no sub-pel search, dist at the full-pel MV
      // (inter_search.cc:650-653)
      res.subpel_dist =
          me_subpel_dist(c, s, b, pic_w, pic_h, pr.p, res.mv_x, res.mv_y);
    } else {
      // SubpelSearch, inter_search.cc:893-949; offsets :38-43
      const int8_t half[9][2] = {{0, 0},  {0, -1}, {0, 1},  {-1, 0}, {1, 0},
                                 {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
      const int8_t qpel[9][2] = {{0, 0},  {0, -1}, {0, 1},  {-1, -1}, {1, -1},
                                 {-1, 0}, {1, 0},  {-1, 1}, {1, 1}};
      uint32_t best_cost = 0xffffffffu, best_dist = 0xffffffffu;
      int best_x = res.mv_x, best_y = res.mv_y;
      for (int pass = 0; pass < 2; pass++) {
        const int base_x = best_x, base_y = best_y;
        const int scale = pass == 0 ? 8 : 4;
        for (int i = pass; i < 9; i++) {
          const int dx = pass == 0 ? half[i][0] : qpel[i][0];
          const int dy = pass == 0 ? half[i][1] : qpel[i][1];
          const int mx = base_x + dx * scale, my = base_y + dy * scale;
          const uint32_t dist =
              me_subpel_dist(c, s, b, pic_w, pic_h, pr.p, mx, my);
          const uint32_t cost =
              dist + ((c.lambda * d_mvd_bits(b.mvp_x, b.mvp_y, mx, my, 0)) >> 16);
          if (cost < best_cost) {
            best_cost = cost;
            best_dist = dist;
            best_x = mx;
            best_y = my;
          }
        }
      }
      res.mv_x = best_x;
      res.mv_y = best_y;
      res.subpel_dist = best_dist;
    }
  }
  if (threadIdx.x == 0) results[bi] = res;
}

#endif  // XVCGPU_K_ME_H_
