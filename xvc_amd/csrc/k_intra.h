// Intra prediction (67-mode set) and the SATD pre-selection pass of the intra
// search: IntraPrediction::ComputeRefSamples /
// FilterRefSamples / Predict (xvc_common_lib/intra_prediction.cc:81-147,
// :342-558, :707-871) and the distortion loop of
// IntraSearch::DetermineSlowIntraModes (xvc_enc_lib/intra_search.cc:189-305).
#ifndef XVCGPU_K_INTRA_H_
#define XVCGPU_K_INTRA_H_

#include "dev_common.h"
#include "k_metric.h"
#include "xvcgpu_internal.h"

__constant__ int8_t kIntraAngle[33] = {-32, -29, -26, -23, -21, -19, -17, -15, -13, -11, -9,
                                       -7,  -5,  -3,  -2,  -1,  0,   1,   2,   3,   5,   7,
                                       9,   11,  13,  15,  17,  19,  21,  23,  26,  29,  32};
__constant__ int16_t kIntraInvAngle[16] = {8192, 4096, 2731, 1638, 1170, 910, 745, 630,
                                           546,  482,  431,  390,  356,  315, 282, 256};
__constant__ int8_t kIntraFilterThr[8] = {0, 20, 20, 14, 2, 0, 20, 0};

// Reference samples of one block, both kinds, each side with the corner in
// front: above[0] = left[0] = corner, above[1 + i], left[1 + i], i < w + h.
struct IntraRefs {
  uint16_t above[2][132];  // [0] unfiltered, [1] filtered
  uint16_t left[2][132];
};

// ComputeRefSamples in closed form (see the oracle for the sequential
// original): every entry is either a reconstructed neighbour sample or the
// nearest available one in the order below-left <- left <- corner <- above <-
// above-right, or the mid value when nothing is available.  All threads of the
// workgroup; ends with a barrier.
__device__ __forceinline__ void intra_build_refs(IntraRefs &r, const xvcgpu_intra_block &b,
                                                 const uint16_t *src, int stride, int bd,
                                                 bool filter, int tid, int nthreads) {
  const int w = b.w, h = b.h, n = w + h;
  const bool has_al = b.neighbors & XVC_INTRA_HAS_ABOVE_LEFT;
  const bool has_a = b.neighbors & XVC_INTRA_HAS_ABOVE;
  const bool has_l = b.neighbors & XVC_INTRA_HAS_LEFT;
  const int ar = b.above_right, bl = b.below_left;
  const int dc = 1 << (bd - 1);
  // value that pads a missing below-left / left side
  int bl_pad;
  if (has_l) bl_pad = src[(ptrdiff_t)(h - 1) * stride - 1];
  else if (has_al) bl_pad = src[-(ptrdiff_t)stride - 1];
  else if (has_a) bl_pad = src[-(ptrdiff_t)stride];
  else bl_pad = dc;
  const int left0 = has_l ? src[-1] : bl_pad;
  const int corner = has_al ? src[-(ptrdiff_t)stride - 1] : left0;
  const int above_last = has_a ? src[-(ptrdiff_t)stride + w - 1] : corner;
  for (int i = tid; i < n; i += nthreads) {
    int l, a;
    if (i < h) {
      l = has_l ? src[(ptrdiff_t)i * stride - 1] : bl_pad;
    } else {
      const int k = i - h;
      l = bl > 0 ? src[(ptrdiff_t)(h + (k < bl ? k : bl - 1)) * stride - 1] : bl_pad;
    }
    if (i < w) {
      a = has_a ? src[-(ptrdiff_t)stride + i] : corner;
    } else {
      const int k = i - w;
      a = ar > 0 ? src[-(ptrdiff_t)stride + w + (k < ar ? k : ar - 1)] : above_last;
    }
    r.left[0][1 + i] = (uint16_t)l;
    r.above[0][1 + i] = (uint16_t)a;
  }
  if (tid == 0) r.above[0][0] = r.left[0][0] = (uint16_t)corner;
  __syncthreads();
  if (!filter) return;
  // FilterRefSamples: [1 2 1] along the L-shaped line, the two ends copied
  for (int i = tid; i < n; i += nthreads) {
    const int pa = i == 0 ? corner : r.above[0][i], pl = i == 0 ? corner : r.left[0][i];
    const int ca = r.above[0][1 + i], cl = r.left[0][1 + i];
    r.above[1][1 + i] = i == n - 1 ? (uint16_t)ca
                                   : (uint16_t)(((ca << 1) + pa + r.above[0][2 + i] + 2) >> 2);
    r.left[1][1 + i] = i == n - 1 ? (uint16_t)cl
                                  : (uint16_t)(((cl << 1) + pl + r.left[0][2 + i] + 2) >> 2);
  }
  if (tid == 0)
    r.above[1][0] = r.left[1][0] =
        (uint16_t)(((corner << 1) + r.above[0][1] + r.left[0][1] + 2) >> 2);
  __syncthreads();
}

__device__ __forceinline__ bool intra_use_filtered(int w, int h, int mode) {
  const int size = ((31 - __clz(w)) + (31 - __clz(h))) >> 1;
  const int dh = mode > 18 ? mode - 18 : 18 - mode, dv = mode > 50 ? mode - 50 : 50 - mode;
  return (dh < dv ? dh : dv) > kIntraFilterThr[size];
}

// Predict for one mode by a group of `nthreads` threads (a wave or a
// workgroup) that share `line` (132 entries) as scratch.  out(y, x) =
// out[y * os + x].  `sync` separates the line-buffer build from its use: a
// no-op for a single wave (LDS operations of a wave complete in order).
template <bool WG_SYNC>
__device__ __forceinline__ void intra_predict(const IntraRefs &r, uint16_t *line, int bd,
                                              bool is_luma, int mode, int w, int h,
                                              uint16_t *out, int os, int tid, int nthreads) {
  const int f = (is_luma && intra_use_filtered(w, h, mode)) ? 1 : 0;
  const bool post = is_luma && w <= 16 && h <= 16;
  const int smax = (1 << bd) - 1;
  if (mode == 0) {  // PlanarPred
    const int wl = 31 - __clz(w), hl = 31 - __clz(h);
    const uint16_t *above = r.above[f] + 1, *left = r.left[f] + 1;
    const int top_right = above[w], bottom_left = left[h];
    const int shift = wl + hl + 1, offset = 1 << (shift - 1);
    for (int p = tid; p < w * h; p += nthreads) {
      const int y = p >> wl, x = p & (w - 1);
      const int hor = (h - 1 - y) * above[x] + (y + 1) * bottom_left;
      const int ver = (w - 1 - x) * left[y] + (x + 1) * top_right;
      out[y * os + x] = (uint16_t)(((hor << wl) + (ver << hl) + offset) >> shift);
    }
    return;
  }
  if (mode == 1) {  // PredIntraDC, unfiltered references
    const uint16_t *above = r.above[0] + 1, *left = r.left[0] + 1;
    int sum = 0;
    for (int i = 0; i < w; i++) sum += above[i];  // every thread: tiny, keeps it barrier-free
    for (int i = 0; i < h; i++) sum += left[i];
    const int total = w + h, dc = (sum + (total >> 1)) / total;
    const int wl = 31 - __clz(w);
    for (int p = tid; p < w * h; p += nthreads) {
      const int y = p >> wl, x = p & (w - 1);
      int v = dc;
      if (post) {
        if (x == 0 && y == 0) v = (above[0] + left[0] + 2 * dc + 2) >> 2;
        else if (y == 0) v = (above[x] + 3 * dc + 2) >> 2;
        else if (x == 0) v = (left[y] + 3 * dc + 2) >> 2;
      }
      out[y * os + x] = (uint16_t)v;
    }
    return;
  }
  // AngularPred: the horizontal half runs on swapped references and writes
  // transposed
  const bool hor = mode < 34;
  const uint16_t *t1 = hor ? r.left[f] : r.above[f];   // rp[i]        = t1[i]
  const uint16_t *t2 = hor ? r.above[f] : r.left[f];   // rp[RS + j]   = t2[1 + j]
  const int pw = hor ? h : w, ph = hor ? w : h;        // size in the swapped frame
  const int angle_offset = hor ? 18 - mode : mode - 50;
  const int angle = kIntraAngle[16 + angle_offset];
  const int pwl = 31 - __clz(pw);
  const uint16_t *ln = t1 + 1;
  if (angle < 0) {
    // project the side edge onto the prediction line
    const int num_projected = -((ph * angle) >> 5) - 1;
    const int inv = kIntraInvAngle[-angle_offset - 1];
    uint16_t *base = line + num_projected + 1;
    for (int i = tid; i < pw + 1 + num_projected; i += nthreads) {
      if (i < pw + 1) {
        base[i - 1] = t1[i];
      } else {
        const int k = i - (pw + 1);
        base[-2 - k] = t2[1 + ((128 + (k + 1) * inv) >> 8) - 1];
      }
    }
    if (WG_SYNC) __syncthreads();
    ln = base;
  }
  const int corner = t1[0];
  for (int p = tid; p < pw * ph; p += nthreads) {
    const int y = p >> pwl, x = p & (pw - 1);
    int v;
    if (angle == 0) {
      v = t1[1 + x];
      if (post && x == 0) v = d_clip3((int)(int16_t)(t1[1] + ((t2[1 + y] - corner) >> 1)), 0, smax);
    } else {
      const int asum = (y + 1) * angle, off = asum >> 5, wt = asum & 31;
      v = wt ? ((32 - wt) * ln[off + x] + wt * ln[off + x + 1] + 16) >> 5 : ln[off + x];
      if (post && x == 0 && (angle == 1 || angle == -1))
        v = d_clip3((int)(int16_t)(v + ((t2[1 + y] - corner) >> 2)), 0, smax);
    }
    if (hor) out[x * os + y] = (uint16_t)v;
    else out[y * os + x] = (uint16_t)v;
  }
  if (WG_SYNC) __syncthreads();  // `line` may be rebuilt by the caller's next mode
}

struct IntraPredShared {
  IntraRefs refs;
  uint16_t line[132];
};

// grid: n jobs; block 256.  One prediction (job.mode) per job, any component,
// written into `pred` at the block's position.
__global__ void __launch_bounds__(256)
intra_pred_kernel(PicView rec, PicView pred, const xvcgpu_intra_block *jobs, int n) {
  __shared__ IntraPredShared s;
  if ((int)blockIdx.x >= n) return;
  const xvcgpu_intra_block b = jobs[blockIdx.x];
  const PlaneView pr = rec.c[b.comp], pp = pred.c[b.comp];
  const bool is_luma = b.comp == 0;
  intra_build_refs(s.refs, b, pr.p + (ptrdiff_t)b.y * pr.stride + b.x, pr.stride, rec.bd,
                   is_luma, threadIdx.x, 256);
  intra_predict<true>(s.refs, s.line, rec.bd, is_luma, b.mode, b.w, b.h,
                      pp.p + (ptrdiff_t)b.y * pp.stride + b.x, pp.stride, threadIdx.x, 256);
}

template <int MS>
struct IntraSatdShared {
  IntraRefs refs;
  uint16_t line[4][132];
  uint16_t orig[MS * MS];
  uint16_t pred[4][MS * MS];
};

// grid: n jobs; block 256 = 4 waves.  Luma: the 67 modes are dealt to the waves;
// a wave predicts into its own LDS tile and takes the SATD against the
// original block.  dist[job * 67 + mode].  MS = largest block side of the batch
// (sizes the LDS tiles, i.e. how many workgroups share a CU); larger jobs are
// skipped.
template <int MS>
__global__ void __launch_bounds__(256)
intra_satd_kernel(PicView orig, PicView rec, const xvcgpu_intra_block *jobs, int n,
                  uint32_t *dist) {
  __shared__ IntraSatdShared<MS> s;
  if ((int)blockIdx.x >= n) return;
  const xvcgpu_intra_block b = jobs[blockIdx.x];
  if (b.w > MS || b.h > MS) return;
  const PlaneView po = orig.c[0], pr = rec.c[0];
  const int w = b.w, h = b.h, wl = 31 - __clz(w);
  for (int p = threadIdx.x; p < w * h; p += 256) {
    const int y = p >> wl, x = p & (w - 1);
    s.orig[y * w + x] = po.p[(ptrdiff_t)(b.y + y) * po.stride + b.x + x];
  }
  intra_build_refs(s.refs, b, pr.p + (ptrdiff_t)b.y * pr.stride + b.x, pr.stride, rec.bd, true,
                   threadIdx.x, 256);  // ends with a barrier: orig is complete too
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int m = wave; m < XVC_INTRA_NUM_MODES; m += 4) {
    intra_predict<false>(s.refs, s.line[wave], rec.bd, true, m, w, h, s.pred[wave], w, lane, 64);
    const uint64_t d = wave_satd(rec.bd, w, h, 0, s.orig, w, s.pred[wave], w);
    if (lane == 0) dist[(size_t)blockIdx.x * XVC_INTRA_NUM_MODES + m] = (uint32_t)d;
  }
}

#endif  // XVCGPU_K_INTRA_H_
