// Intra prediction (67-mode set) and the SATD pre-selection pass of the intra
// search: IntraPrediction::ComputeRefSamples /
// FilterRefSamples / Predict (xvc_common_lib/intra_prediction.cc:81-147,
// :342-558, :707-871) and the distortion loop of
// IntraSearch::DetermineSlowIntraModes (xvc_enc_lib/intra_search.cc:189-305).
#ifndef XVCGPU_K_INTRA_H_
#define XVCGPU_K_INTRA_H_

#include "dev_common.h"
#include "k_metric.h"
#include "k_tx2.h"
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
// group (a workgroup with WG_SYNC, else one wave - whose LDS operations complete
// in order, so no barrier is needed); the entries are complete on return.
template <bool WG_SYNC>
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
  if (WG_SYNC) __syncthreads();
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
  if (WG_SYNC) __syncthreads();
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

// PredLmChroma (intra_prediction.cc:560-585), 4:2:0, by one workgroup of 256:
// RescaleLuma (:873-906) of the CU's reconstructed luma plus the row above /
// column to the left into LDS, DeriveLmParams (:587-686: the neighbour sums by
// the first wave, the integer model by one thread), AddLinearModel.
struct IntraLmShared {
  uint16_t sub[33 * 33];  // (h + 1) x (w + 1), block origin at [1][1]
  int scale, shift, offset;
};

__device__ __forceinline__ int d_log2_floor(int x) { return x > 1 ? 31 - __clz(x) : 0; }

__device__ __forceinline__ void intra_lm_chroma(IntraLmShared &s, const xvcgpu_intra_block &b,
                                                const PlaneView &luma, const PlaneView &chroma,
                                                int bd, uint16_t *out, int os) {
  const int w = b.w, h = b.h, tid = threadIdx.x;
  const bool has_above = b.y > 0, has_left = b.x > 0;
  constexpr int SS = 33;
  uint16_t *sub = s.sub + SS + 1;
  const int ls = luma.stride;
  const uint16_t *src0 = luma.p + (ptrdiff_t)(2 * b.y) * ls + 2 * b.x;
  const int x0 = has_left ? -1 : 0, y0 = has_above ? -1 : 0;
  const int cols = w - x0, rows = h - y0;
  for (int p = tid; p < cols * rows; p += 256) {
    const int yy = y0 + p / cols, xx = x0 + p % cols;
    const uint16_t *r = src0 + (ptrdiff_t)(2 * yy) * ls;
    int v;
    if (xx == 0 && !has_left) {
      v = (r[0] + r[ls] + 1) >> 1;
    } else {  // column -1 uses the same taps at 2 * (-1) = -2: src[-3], src[-2], src[-1]
      const uint16_t *q = r + 2 * xx;
      v = (q[-1] + 2 * q[0] + q[1] + q[-1 + ls] + 2 * q[ls] + q[1 + ls] + 4) >> 3;
    }
    sub[yy * SS + xx] = (uint16_t)v;
  }
  __syncthreads();
  if (tid < 64) {
    int sx = 0, sy = 0, sxx = 0, sxy = 0, nbr = 0;
    if (has_above || has_left) {
      const uint16_t *cb = chroma.p + (ptrdiff_t)b.y * chroma.stride + b.x;
      const int cs = chroma.stride;
      const int dx = has_left ? (w / h > 1 ? w / h : 1) : 1;
      const int dy = has_above ? (h / w > 1 ? h / w : 1) : 1;
      const int na = has_above ? w / dx : 0, nl = has_left ? h / dy : 0;
      nbr = na + nl;
      for (int i = tid; i < nbr; i += 64) {  // nbr <= 64: one step
        int r, c;
        if (i < na) {
          r = sub[-SS + i * dx];
          c = cb[-cs + i * dx];
        } else {
          const int yy = (i - na) * dy;
          r = sub[yy * SS - 1];
          c = cb[(ptrdiff_t)yy * cs - 1];
        }
        sx += r; sy += c; sxx += r * r; sxy += r * c;
      }
      sx = group_sum<64>(sx);
      sy = group_sum<64>(sy);
      sxx = group_sum<64>(sxx);
      sxy = group_sum<64>(sxy);
    }
    if (tid == 0) {
      int scale = 0, shift = 0, offset = 1 << (bd - 1);
      if (nbr > 0) {
        int size_shift = 1;
        while ((1 << size_shift) < nbr) size_shift++;
        if (size_shift > 15 - bd) {
          const int sh = size_shift + bd - 15;
          sx = (sx + (1 << (sh - 1))) >> sh;
          sy = (sy + (1 << (sh - 1))) >> sh;
          sxx = (sxx + (1 << (sh - 1))) >> sh;
          sxy = (sxy + (1 << (sh - 1))) >> sh;
          size_shift -= sh;
        }
        const int avg_x = sx >> size_shift, avg_y = sy >> size_shift;
        const int x_frac = sx & ((1 << size_shift) - 1), y_frac = sy & ((1 << size_shift) - 1);
        const int vxy = sxy - ((avg_x * avg_y) << size_shift) - avg_x * y_frac - avg_y * x_frac;
        const int vxx = sxx - ((avg_x * avg_x) << size_shift) - 2 * avg_x * x_frac;
        int shift_xy = vxy == 0 ? 0 : d_log2_floor(d_abs(vxy)) - bd + 2;
        shift_xy = shift_xy < 0 ? 0 : shift_xy;
        int shift_xx = vxx == 0 ? 0 : d_log2_floor(d_abs(vxx)) - 5;
        shift_xx = shift_xx < 0 ? 0 : shift_xx;
        const int vxy_s = vxy >> shift_xy, vxx_s = vxx >> shift_xx;
        const int total_shift = bd + shift_xx + 4 + 7 - 13 - shift_xy;
        if (vxx_s < 32) {
          offset = avg_y;
        } else {
          int sc = (int)((uint32_t)vxy_s * (uint32_t)(((1 << (bd + 4)) + (vxx_s / 2)) / vxx_s));
          sc >>= total_shift;
          sc = d_clip3(sc, -256, 255);
          scale = 128 * sc;
          const int base_shift =
              d_log2_floor(d_abs(scale) + (scale < 0 ? -1 : 0)) - (scale ? 5 : 0);
          shift = 13 - base_shift;
          scale >>= base_shift;
          offset = avg_y - ((scale * avg_x) >> shift);
        }
      }
      s.scale = scale;
      s.shift = shift;
      s.offset = offset;
    }
  }
  __syncthreads();
  const int scale = s.scale, shift = s.shift, offset = s.offset, smax = (1 << bd) - 1;
  for (int p = tid; p < w * h; p += 256) {
    const int yy = p / w, xx = p - yy * w;
    out[yy * os + xx] = (uint16_t)d_clip3(((scale * sub[yy * SS + xx]) >> shift) + offset, 0, smax);
  }
}

struct IntraPredShared {
  IntraRefs refs;
  uint16_t line[132];
  IntraLmShared lm;
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
  if (b.mode == XVC_INTRA_MODE_LM_CHROMA) {
    if (!is_luma && b.w <= 32 && b.h <= 32)
      intra_lm_chroma(s.lm, b, rec.c[0], pr, rec.bd, pp.p + (ptrdiff_t)b.y * pp.stride + b.x,
                      pp.stride);
    return;
  }
  intra_build_refs<true>(s.refs, b, pr.p + (ptrdiff_t)b.y * pr.stride + b.x, pr.stride,
                         rec.bd, is_luma, threadIdx.x, 256);
  intra_predict<true>(s.refs, s.line, rec.bd, is_luma, b.mode, b.w, b.h,
                      pp.p + (ptrdiff_t)b.y * pp.stride + b.x, pp.stride, threadIdx.x, 256);
}

template <int MS>
struct alignas(16) IntraSatdShared {
  IntraRefs refs;
  uint16_t line[4][MS <= 16 ? 8 : 1][132];  // small blocks: up to 8 modes per wave
  uint16_t orig[MS * MS];
  uint16_t orig_t[MS <= 16 ? MS * MS : 8];  // square 8 / 16 blocks: the original transposed
  uint16_t pred[4][MS <= 16 ? 512 : MS * MS];
};

// ---- square 8x8 / 16x16 blocks: prediction straight into the SATD's registers ----
// A lane owns one row of one 8x8 tile (8 lanes = a tile, 32 lanes = a 16x16
// block) and computes the 8 predicted samples of that row itself - no
// prediction tile in LDS, no per-sample index arithmetic: along a row of an
// angular mode the offset and the weight are constants and the 8 samples come
// from 9 consecutive reference samples.  The horizontal half of the modes
// (2 .. 33) is the vertical half on swapped references producing the transposed
// block: its lanes compare with the TRANSPOSED original instead - the sum of
// |H d H^T| over the tiles does not change when d is transposed.
template <int CTRL>
__device__ __forceinline__ int intra_dpp(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
#define INTRA_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define INTRA_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define INTRA_DPP_HALF_MIRROR 0x141 // lane i <-> 7 - i inside each 8 lanes
#define INTRA_DPP_ROR8 0x128        // row_ror:8: lane i <-> i ^ 8 inside each 16 lanes

// SATD of an 8x8 tile whose row `row` (= lane & 7) this lane holds as 8
// differences: rows butterflied in registers, columns across the 8 lanes with
// DPP moves.  The three lane pairings are i^1, i^2 and 7-i (= i^7: what DPP
// offers for 8 lanes); which lane of a pair keeps the sum is chosen so that the
// partners of the later stages hold the same kind of term (stage 1: bit0^bit2,
// stage 2: bit1^bit2, stage 3: bit2) - a Hadamard transform up to the order and
// the signs of its outputs, which the sum of magnitudes does not see.
// Returns the tile's (sum + 2) >> 2 in all 8 lanes.
__device__ __forceinline__ int intra_satd8_rows(int m[8], int row) {
#pragma unroll
  for (int len = 1; len < 8; len <<= 1)
#pragma unroll
    for (int i = 0; i < 8; i += len << 1)
#pragma unroll
      for (int j = i; j < i + len; j++) {
        const int u = m[j], v = m[j + len];
        m[j] = u + v;
        m[j + len] = u - v;
      }
  const int s1 = ((row ^ (row >> 2)) & 1) ? -1 : 1;
  const int s2 = (((row >> 1) ^ (row >> 2)) & 1) ? -1 : 1;
  const int s3 = (row & 4) ? -1 : 1;
#pragma unroll
  for (int x = 0; x < 8; x++) m[x] = intra_dpp<INTRA_DPP_XOR1>(m[x]) + s1 * m[x];
#pragma unroll
  for (int x = 0; x < 8; x++) m[x] = intra_dpp<INTRA_DPP_XOR2>(m[x]) + s2 * m[x];
  int sum = 0;
#pragma unroll
  for (int x = 0; x < 8; x++) sum += d_abs(intra_dpp<INTRA_DPP_HALF_MIRROR>(m[x]) + s3 * m[x]);
  sum += intra_dpp<INTRA_DPP_XOR1>(sum);
  sum += intra_dpp<INTRA_DPP_XOR2>(sum);
  sum += intra_dpp<INTRA_DPP_HALF_MIRROR>(sum);
  return (sum + 2) >> 2;
}

// Row y, columns tx .. tx+7 of the W x W luma prediction of `mode` in the
// mode's own frame (modes 2 .. 33: the swapped one, i.e. column y of the block).
// `line`: the group's scratch for the projected reference line of the negative
// angles, built here by the group's `gl` lanes (lane index `sub`).  dc: the
// block's DC value.  Same arithmetic as intra_predict.
template <int W>
__device__ __forceinline__ void intra_row8(const IntraRefs &r, uint16_t *line, int bd, int mode,
                                           int y, int tx, int dc, int sub, int gl, int v[8]) {
  const int f = intra_use_filtered(W, W, mode) ? 1 : 0;
  const int smax = (1 << bd) - 1;
  constexpr int WL = W == 16 ? 4 : 3;
  if (mode == 0) {  // PlanarPred
    const uint16_t *above = r.above[f] + 1, *left = r.left[f] + 1;
    const int top_right = above[W], bottom_left = left[W], ly = left[y];
    constexpr int shift = 2 * WL + 1, offset = 1 << (shift - 1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int x = tx + k;
      const int hor = (W - 1 - y) * above[x] + (y + 1) * bottom_left;
      const int ver = (W - 1 - x) * ly + (x + 1) * top_right;
      v[k] = ((hor << WL) + (ver << WL) + offset) >> shift;
    }
    return;
  }
  if (mode == 1) {  // PredIntraDC with its edge filter (blocks up to 16x16)
    const uint16_t *above = r.above[0] + 1, *left = r.left[0] + 1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int x = tx + k;
      v[k] = y == 0 ? (above[x] + 3 * dc + 2) >> 2 : dc;
    }
    if (tx == 0) v[0] = y == 0 ? (above[0] + left[0] + 2 * dc + 2) >> 2 : (left[y] + 3 * dc + 2) >> 2;
    return;
  }
  const bool hor = mode < 34;
  const uint16_t *t1 = hor ? r.left[f] : r.above[f];
  const uint16_t *t2 = hor ? r.above[f] : r.left[f];
  const int angle_offset = hor ? 18 - mode : mode - 50;
  const int angle = kIntraAngle[16 + angle_offset];
  const uint16_t *ln = t1 + 1;
  if (angle < 0) {
    const int num_projected = -((W * angle) >> 5) - 1;
    const int inv = kIntraInvAngle[-angle_offset - 1];
    uint16_t *base = line + num_projected + 1;
    for (int i = sub; i < W + 1 + num_projected; i += gl) {
      if (i < W + 1) {
        base[i - 1] = t1[i];
      } else {
        const int k = i - (W + 1);
        base[-2 - k] = t2[1 + ((128 + (k + 1) * inv) >> 8) - 1];
      }
    }
    wave_sync();
    ln = base;
  }
  const int corner = t1[0];
  if (angle == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = t1[1 + tx + k];
    if (tx == 0) v[0] = d_clip3((int)(int16_t)(t1[1] + ((t2[1 + y] - corner) >> 1)), 0, smax);
    return;
  }
  const int asum = (y + 1) * angle, off = asum >> 5, wt = asum & 31;
  const uint16_t *a = ln + off + tx;
  int prev = a[0];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int next = a[k + 1];
    v[k] = ((32 - wt) * prev + wt * next + 16) >> 5;
    prev = next;
  }
  if (tx == 0 && (angle == 1 || angle == -1))
    v[0] = d_clip3((int)(int16_t)(v[0] + ((t2[1 + y] - corner) >> 2)), 0, smax);
}

// grid: (n jobs, S); block 256 = 4 waves.  Luma: the 67 modes are dealt to the
// 4 * S waves that work on a job (S = 1 for big batches; up to 17 - one mode
// per wave - when the batch alone cannot fill the chip, e.g. one anti-diagonal
// of a picture); a wave predicts into its own LDS tile and takes the SATD
// against the original block.  dist[job * 67 + mode].  MS = largest block side of the batch
// (sizes the LDS tiles, i.e. how many workgroups share a CU); larger jobs are
// skipped.
template <int MS>
__global__ void __launch_bounds__(256)
intra_satd_kernel(PicView orig, PicView rec, const xvcgpu_intra_block *jobs, int n,
                  uint32_t *dist) {
  __shared__ IntraSatdShared<MS> s;
  if ((int)blockIdx.x >= n) return;
  const xvcgpu_intra_block b = jobs[blockIdx.x];
  if (b.w > MS || b.h > MS) {
    // larger than the caller's max_block_size: every mode's distortion all ones
    // (XVCGPU_INTRA_SATD_UNSUPPORTED) instead of stale memory
    if (blockIdx.y == 0)
      for (int m = threadIdx.x; m < XVC_INTRA_NUM_MODES; m += blockDim.x)
        dist[(size_t)blockIdx.x * XVC_INTRA_NUM_MODES + m] = 0xffffffffu;
    return;
  }
  const PlaneView po = orig.c[0], pr = rec.c[0];
  const int w = b.w, h = b.h, wl = 31 - __clz(w);
  const bool rows_path = MS <= 16 && w == h && (w == 8 || w == 16);
  for (int p = threadIdx.x; p < w * h; p += 256) {
    const int y = p >> wl, x = p & (w - 1);
    const uint16_t o = po.p[(ptrdiff_t)(b.y + y) * po.stride + b.x + x];
    s.orig[y * w + x] = o;
    if (MS <= 16 && rows_path) s.orig_t[x * w + y] = o;
  }
  intra_build_refs<true>(s.refs, b, pr.p + (ptrdiff_t)b.y * pr.stride + b.x, pr.stride, rec.bd,
                         true, threadIdx.x, 256);  // ends with a barrier: orig is complete too
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (MS <= 16 && rows_path) {
    // K modes side by side in one wave: K = 2 (16x16: four tiles of 8 lanes per
    // mode) or 8 (8x8: one tile); lane = (mode slot, tile, row of the tile)
    const int K = w == 16 ? 2 : 8, gl = 64 / K, g = lane / gl, sub = lane - g * gl;
    const int row = lane & 7, q = w == 16 ? (lane >> 3) & 3 : 0;
    const int tx = (q & 1) * 8, y = (q >> 1) * 8 + row;
    // the lane's row of the original, and of the transposed original
    int od[2][8];
    {
      const uint4 a = *reinterpret_cast<const uint4 *>(s.orig + y * w + tx);
      const uint4 t = *reinterpret_cast<const uint4 *>(s.orig_t + y * w + tx);
      const uint32_t ua[4] = {a.x, a.y, a.z, a.w}, ut[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        od[0][2 * k] = ua[k] & 0xffff;
        od[0][2 * k + 1] = ua[k] >> 16;
        od[1][2 * k] = ut[k] & 0xffff;
        od[1][2 * k + 1] = ut[k] >> 16;
      }
    }
    // PredIntraDC's value, by the wave (unfiltered references)
    int dc;
    {
      int t = lane < w ? s.refs.above[0][1 + lane] : (lane < 2 * w ? s.refs.left[0][1 + lane - w] : 0);
      t = group_sum<64>(t);
      dc = (t + w) / (2 * w);
    }
    for (int m0 = (blockIdx.y * 4 + wave) * K; m0 < XVC_INTRA_NUM_MODES;
         m0 += 4 * gridDim.y * K) {
      const int m = m0 + g < XVC_INTRA_NUM_MODES ? m0 + g : XVC_INTRA_NUM_MODES - 1;
      int v[8];
      if (w == 16) intra_row8<16>(s.refs, s.line[wave][g], rec.bd, m, y, tx, dc, sub, gl, v);
      else intra_row8<8>(s.refs, s.line[wave][g], rec.bd, m, y, tx, dc, sub, gl, v);
      const int t = (m >= 2 && m < 34) ? 1 : 0;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = od[t][k] - v[k];
      int d = intra_satd8_rows(v, row);
      if (w == 16) {
        d += intra_dpp<INTRA_DPP_ROR8>(d);
        d += __shfl_xor(d, 16, XVC_WAVE);
      }
      if (sub == 0 && m0 + g < XVC_INTRA_NUM_MODES)
        dist[(size_t)blockIdx.x * XVC_INTRA_NUM_MODES + m] = (uint32_t)(d >> (rec.bd - 8));
      wave_sync();   // the group's line buffer is rebuilt by its next mode
    }
    return;
  }
  for (int m = blockIdx.y * 4 + wave; m < XVC_INTRA_NUM_MODES; m += 4 * gridDim.y) {
    intra_predict<false>(s.refs, s.line[wave][0], rec.bd, true, m, w, h, s.pred[wave], w, lane, 64);
    const uint64_t d = wave_satd(rec.bd, w, h, 0, s.orig, w, s.pred[wave], w);
    if (lane == 0) dist[(size_t)blockIdx.x * XVC_INTRA_NUM_MODES + m] = (uint32_t)d;
  }
}

// ---- mode selection on the device --------------------------------------------
// grid: ceil(n / 4); block 256: wave per CU.  mode = arg min over the 67
// entries of dist[cu] (+ cost[cu], the caller's rate term, when given); first
// minimum wins.  The choice is written to modes[cu] and - when given - into the
// `per_cu` prediction jobs of the CU (all components use it: DM chroma) and
// into the coefficient-scan bits of its `per_cu` transform blocks
// (TransformHelper::DetermineScanOrder, transform.cc:1614-1637: CUs below
// 16x16, horizontal scan within 10 modes of vertical, vertical scan within 10
// of horizontal).
__global__ void __launch_bounds__(256)
intra_select_kernel(const uint32_t *dist, const uint32_t *cost, int n, int32_t *modes,
                    xvcgpu_intra_block *jobs, xvcgpu_tx_block *blocks, int per_cu) {
  // one wave per CU: two entries per lane, keyed wave minimum (value, mode)
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const uint32_t *d = dist + (size_t)i * XVC_INTRA_NUM_MODES;
  const uint32_t *c = cost ? cost + (size_t)i * XVC_INTRA_NUM_MODES : nullptr;
  unsigned long long key = ~0ull;
#pragma unroll
  for (int k = lane; k < XVC_INTRA_NUM_MODES; k += 64) {
    const unsigned long long v = (unsigned long long)d[k] + (c ? c[k] : 0u);
    const unsigned long long kk = (v << 7) | (unsigned)k;   // first minimum wins
    key = kk < key ? kk : key;
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const unsigned long long o = __shfl_xor(key, s, XVC_WAVE);
    key = o < key ? o : key;
  }
  const int m = (int)(key & 127);
  if (lane == 0 && modes) modes[i] = m;
  if (lane < per_cu) {
    const int k = lane;
    if (jobs) jobs[(size_t)i * per_cu + k].mode = (uint8_t)m;
    if (blocks) {
      xvcgpu_tx_block &t = blocks[(size_t)i * per_cu + k];
      // the CU's luma size decides (block 0 of the CU is its luma block)
      const xvcgpu_tx_block &l = blocks[(size_t)i * per_cu];
      int scan = 0;
      if (l.w < 16 && l.h < 16) {
        const int dv = m > 50 ? m - 50 : 50 - m, dh = m > 18 ? m - 18 : 18 - m;
        scan = dv < 10 ? 1 : (dh < 10 ? 2 : 0);
      }
      t.intra_pic = (uint8_t)((t.intra_pic & ~(3 << XVC_TXF_SCAN_SHIFT)) |
                              (scan << XVC_TXF_SCAN_SHIFT));
    }
  }
}

// ---- prediction + TransformAndReconstruct fused -----------------------------
struct IntraWaveShared {
  IntraRefs refs;
  uint16_t line[132];
  uint16_t pred[16 * 16];
};

// grid: ceil(n / TX2_WAVES); block: TX2_WAVES waves, one job per wave: the
// block's prediction (jobs[i]) goes to an LDS tile and straight into the
// one-wave residual pipeline of blocks[i] (k_tx2.h): MODE = TX_MODE_FULL
// (encoder: levels / nnz written) or TX_MODE_INV (decoder: levels / nnz read).
// Blocks up to 16x16 (tx_small_job); others are left to the unfused entry
// points.  In-place on `rec`: the jobs of one batch are mutually independent.
template <int MODE>
__global__ void __launch_bounds__(64 * TX2_WAVES)
intra_recon_wave_kernel(PicView orig, PicView rec, const xvcgpu_intra_block *jobs,
                        const xvcgpu_tx_block *blocks, int n, int16_t *levels,
                        const uint32_t *level_off, int32_t *nnz_out,
                        const int16_t *tx_tables, const int16_t *tx_tables_t,
                        TxTableLayout lay) {
  __shared__ Tx2Shared s_all[TX2_WAVES];
  __shared__ IntraWaveShared i_all[TX2_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bi = blockIdx.x * TX2_WAVES + wave;
  if (bi >= n) return;
  const xvcgpu_tx_block b = blocks[bi];
  if (!tx_small_job(b)) return;
  const xvcgpu_intra_block j = jobs[bi];
  IntraWaveShared &iw = i_all[wave];
  const PlaneView pr = rec.c[b.comp];
  const bool is_luma = b.comp == 0;
  intra_build_refs<false>(iw.refs, j, pr.p + (ptrdiff_t)j.y * pr.stride + j.x, pr.stride,
                          rec.bd, is_luma, lane, 64);
  intra_predict<false>(iw.refs, iw.line, rec.bd, is_luma, j.mode, j.w, j.h, iw.pred, j.w,
                       lane, 64);
  tx2_job<MODE>(s_all[wave], b, bi, rec.bd, orig.c[b.comp], iw.pred, j.w, pr, levels,
                level_off, nnz_out, tx_tables, tx_tables_t, lay);
}

#endif  // XVCGPU_K_INTRA_H_
