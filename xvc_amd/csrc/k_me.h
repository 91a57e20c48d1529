// k_me.h -- shared pieces of the motion-estimation kernels (k_me2.h):
// per-job context, the rate term of the full-pel cost, window tests.
//
// Exactness argument used by the kernels:
// every selection in the reference is a left-to-right fold
// `if (cost < best) best = cost` with
// cost = dist + ((lambda16*bits) >> 16) >= dist, so the early
// `dist >= best -> skip` (inter_tz_search.cc:263) never changes the outcome
// and candidate costs do not depend on the running state: phases evaluate all
// their candidates in parallel and then replay the ordered fold (or take the
// minimum with lowest-index tie-break, which is the same thing).
#ifndef XVCGPU_K_ME_H_
#define XVCGPU_K_ME_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

// 2-byte-aligned 16 / 8 byte loads (candidate blocks sit at arbitrary
// full-pel offsets); gfx950 executes them as one global_load_dwordx4 / x2.
struct __attribute__((packed, aligned(2))) U16x8 {
  uint32_t v[4];
};
struct __attribute__((packed, aligned(2))) U16x4 {
  uint32_t v[2];
};

// Copy `rows` rows of `cpr` 8-sample chunks from global memory (row stride
// ss samples, any 2-byte alignment) to LDS (row stride ds samples, 16-byte
// aligned rows) by one wave.  The loads of a batch (up to 8 per lane) are all
// issued before the first store, so the copy costs one memory round trip per
// batch (B loads per lane) rather than one per loop iteration.
template <int B = 8>
__device__ __forceinline__ void wave_copy_chunks(uint16_t *dst, int ds, const uint16_t *src,
                                                 int ss, int rows, int cpr) {
  const int lane = threadIdx.x & 63;
  const int n = rows * cpr;
  const uint32_t inv = (65536u + (uint32_t)cpr - 1u) / (uint32_t)cpr;  // exact for n < 4096
  for (int base = lane; base < n; base += 64 * B) {
    U16x8 v[B];
    int off[B];
#pragma unroll
    for (int u = 0; u < B; u++) {
      const int i = base + 64 * u;
      const int r = (int)(((uint32_t)i * inv) >> 16), ch = i - r * cpr;
      off[u] = r * ds + ch * 8;
      if (i < n) v[u] = *reinterpret_cast<const U16x8 *>(src + (ptrdiff_t)r * ss + ch * 8);
    }
#pragma unroll
    for (int u = 0; u < B; u++)
      if (base + 64 * u < n)
        *reinterpret_cast<uint4 *>(dst + off[u]) =
            make_uint4(v[u].v[0], v[u].v[1], v[u].v[2], v[u].v[3]);
  }
}

struct MeCtx {
  int bd, w, h, rows, row_step, sad_shift, sad_mul;  // kSad / kSadFast
  const uint16_t *ref;  // reference plane at the CU position
  int rs;
  int mvp_x, mvp_y, down;
  uint32_t lambda;
  int min_x, min_y, max_x, max_y;  // TZ window (mv_min / mv_max)
  // CUs that try local illumination compensation: kSadAcOnly[Fast]
  // (GetFullpelMetric, inter_search.cc:1059-1069).  orig_sum = sum of the
  // original block over the rows the metric visits.
  bool ac;
  int orig_sum;
};

// CheckCostBest's cost (inter_tz_search.cc:261-276).
__device__ __forceinline__ uint32_t me_cost(const MeCtx &c, uint32_t dist,
                                            int mx, int my) {
  const uint32_t bits = d_mvd_bits_fullpel(c.mvp_x, c.mvp_y, mx, my, c.down);
  return dist + ((c.lambda * bits) >> 16);
}

enum { TZ_LEFT = TZP_LEFT, TZ_RIGHT = TZP_RIGHT, TZ_UP = TZP_UP, TZ_DOWN = TZP_DOWN };

// IsInside<Dir> (inter_tz_search.cc:278-302): a single bound per direction.
__device__ __forceinline__ bool tz_inside(const MeCtx &c, int dir, int mx, int my) {
  switch (dir) {
    case TZ_UP: return my >= c.min_y;
    case TZ_DOWN: return my <= c.max_y;
    case TZ_LEFT: return mx >= c.min_x;
    default: return mx <= c.max_x;
  }
}

struct TzState {  // SearchState, inter_tz_search.cc:66-82
  int bx, by;
  uint32_t cost;
  int last_pos, last_range;
};

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

#endif  // XVCGPU_K_ME_H_
