// k_bipred.h -- bi-prediction: M2 / T2 / T7 (the refinement search on the
// int16 target 2*orig - pred_other) and I2 (two 14-bit predictions + AddAvg).
//
//   bipred_search_kernel  = one inner step of InterSearch::SearchBiIterative
//     (inter_search.cc:392-433): MotionCompensation of the other list,
//     SubtractWeighted (sample_buffer.h:147-161), then
//     MotionEstNormal<int16 target>(kFullSearch, bipred) (:606-662) =
//     FullSearch +-4 (:853-891) + SubpelSearch (:893-964) on that target.
//   mc_bipred_kernel      = InterPrediction::MotionCompensation for a bi-pred
//     CU (inter_prediction.cc:710-738): MotionCompUniPred -> int16 per list
//     (:1156-1172, Filter*Bipred) and AddAvgBi (:1545-1547).
//
// One workgroup of BI_WAVES(MS) waves per job.  The candidates of a phase (81
// full-pel positions, 9 + 8 sub-pel positions) are dealt round-robin to the
// waves; every candidate cost is independent of the running best (see
// k_me.h), so the ordered strict-< fold of the reference equals the minimum
// of (cost << 8 | index).
#ifndef XVCGPU_K_BIPRED_H_
#define XVCGPU_K_BIPRED_H_

#include "dev_common.h"
#include "dev_tables.h"
#include "k_interp.h"
#include "k_me.h"
#include "k_metric.h"
#include "k_recon.h"
#include "xvcgpu_internal.h"

// Waves per workgroup (= job).  The 64 class holds 17 KB of interpolation scratch per wave.
// Round 6: 8 waves for the 32 / 64 classes (3 / 4 before: a 64x64 job took 88 us, the longest
// launch of an engine round; 146 KB of LDS, one job per CU - these launches carry a few
// jobs each).  Engine at k = 16: 0.72 -> 0.85 pictures/s, chained walk 109 -> 101 us / state.
#ifndef BI_WAVES_64
#define BI_WAVES_64 8
#endif
#ifndef BI_WAVES_32
#define BI_WAVES_32 8
#endif
#define BI_WAVES(MS) ((MS) > 32 ? BI_WAVES_64 : ((MS) > 16 ? BI_WAVES_32 : 4))

template <int MS>
struct __attribute__((aligned(16))) BiShared {
  int16_t target[MS * MS];  // row stride w
  struct {
    int16_t tmp[MS * (MS + 7)];
    uint16_t pred[MS * MS];
  } wv[BI_WAVES(MS)];
  unsigned long long key[BI_WAVES(MS)];
  uint32_t dist[18];
};

// MotionCompensationMv of one luma block into LDS (clip, split, interpolate).
__device__ __forceinline__ void bi_mc_luma(int bd, const xvcgpu_me_block &b,
                                           const PlaneView &pr, int mx, int my,
                                           int16_t *tmp, uint16_t *pred) {
  d_clip_mv(b.x, b.y, pr.w, pr.h, mx, my);
  const uint16_t *r = pr.p + (ptrdiff_t)(b.y + (my >> 4)) * pr.stride + b.x + (mx >> 4);
  wave_interp_block<false>(bd, b.w, b.h, mx & 15, my & 15, r, pr.stride, tmp, pred);
}

// The same by the waves of a workgroup: wave k the rows [k * h / n, (k + 1) * h / n) of the
// block, n = min(waves, h / 4) (sample for sample what one wave computes; a slab's
// horizontal pass covers its own 7 extra rows).  pred: the whole block's, tmp: the wave's.
__device__ __forceinline__ void bi_mc_luma_slabs(int bd, const xvcgpu_me_block &b,
                                                 const PlaneView &pr, int mx, int my,
                                                 int16_t *tmp, uint16_t *pred, int wave, int nw) {
  const int n = (b.h >> 2) < nw ? (b.h >> 2) : nw;
  if (wave >= n) return;
  const int rows = b.h / n, row0 = wave * rows;
  d_clip_mv(b.x, b.y, pr.w, pr.h, mx, my);
  const uint16_t *r = pr.p + (ptrdiff_t)(b.y + (my >> 4) + row0) * pr.stride + b.x + (mx >> 4);
  wave_interp_block<false>(bd, b.w, rows, mx & 15, my & 15, r, pr.stride, tmp, pred + row0 * b.w);
}

// DeriveLicParams (inter_prediction.cc:1577-1663) by the first wave of the
// workgroup; scale / offset are left in *s_scale / *s_offset (LDS) and are
// visible after the next __syncthreads().  mx, my: the CU's clipped vector.
__device__ __forceinline__ void wg_lic_model(int bd, int comp, int bx, int by, int bw, int bh,
                                             int mx, int my, int neighbors, int above_x,
                                             int above_y, int left_x, int left_y, int pic_w,
                                             int pic_h, const PlaneView &pr, const PlaneView &pc,
                                             int *s_scale, int *s_offset) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  const int cs = comp ? 1 : 0, shift = 4 + cs;
  const int cx = bx >> cs, cy = by >> cs, cw = bw >> cs, ch = bh >> cs;
  const bool has_above = neighbors & XVC_LIC_HAS_ABOVE, has_left = neighbors & XVC_LIC_HAS_LEFT;
  const int full_x = (mx + (1 << (shift - 1))) >> shift, full_y = (my + (1 << (shift - 1))) >> shift;
  const int step = (cw < ch ? cw : ch) > 8 ? 2 : 1;
  const int dx = step * (cw / ch > 1 ? cw / ch : 1), dy = step * (ch / cw > 1 ? ch / cw : 1);
  const int na = has_above ? cw / dx : 0, nl = has_left ? ch / dy : 0;
  const int nbr = na + nl;
  const uint16_t *rb = pr.p + (ptrdiff_t)cy * pr.stride + cx;
  const uint16_t *sb = pc.p + (ptrdiff_t)cy * pc.stride + cx;
  int sx = 0, sy = 0, sxx = 0, sxy = 0;
  for (int i = lane; i < nbr; i += 64) {
    int a, d;
    if (i < na) {
      int vx = full_x, vy = full_y;
      d_clip_mv(above_x, above_y, pic_w, pic_h, vx, vy);
      a = rb[(ptrdiff_t)(vy - 1) * pr.stride + vx + i * dx];
      d = sb[-(ptrdiff_t)pc.stride + i * dx];
    } else {
      int vx = full_x, vy = full_y;
      d_clip_mv(left_x, left_y, pic_w, pic_h, vx, vy);
      const int yy = (i - na) * dy;
      a = rb[(ptrdiff_t)(vy + yy) * pr.stride + vx - 1];
      d = sb[(ptrdiff_t)yy * pc.stride - 1];
    }
    sx += a; sy += d; sxx += a * a; sxy += a * d;
  }
  sx = group_sum<64>(sx);
  sy = group_sum<64>(sy);
  sxx = group_sum<64>(sxx);
  sxy = group_sum<64>(sxy);
  if (lane != 0) return;
  int scale = 32, offset = 0;
  if (nbr > 0) {
    int size_shift = 1;
    while ((1 << size_shift) < nbr) size_shift++;
    int base_shift = bd + size_shift - 15;
    base_shift = base_shift < 0 ? 0 : base_shift;
    const int avg_x = sx >> base_shift, avg_y = sy >> base_shift;
    const int xx_offset = sxx >> 7;
    const int avg_xy = ((sxy + xx_offset) >> (2 * base_shift)) << size_shift;
    const int avg_xx = ((sxx + xx_offset) >> (2 * base_shift)) << size_shift;
    const int vxy = avg_xy - avg_x * avg_y, vxx = avg_xx - avg_x * avg_x;
    const int msb = vxx == 0 ? 0 : 32 - __clz(d_abs(vxx));
    int shift_xx = msb - 6;
    shift_xx = shift_xx < 0 ? 0 : shift_xx;
    int shift_xy = shift_xx - 12;
    shift_xy = shift_xy < 0 ? 0 : shift_xy;
    const int total_shift = 15 - 5 + shift_xx - shift_xy;
    const int vxy_s = vxy >> shift_xy;
    const int vxx_s = d_clip3(vxx >> shift_xx, 0, 63);
    if (vxx_s != 0) {
      const int vxx_scaled = ((1 << 15) + (vxx_s / 2)) / vxx_s;
      const int sc = (int)((long long)vxy_s * vxx_scaled) >> total_shift;
      scale = d_clip3(sc, 0, 128);
      const int off = (sy - ((scale * sx) >> 5) + (1 << (size_shift - 1))) >> size_shift;
      offset = d_clip3(off, -(1 << (bd - 1)), (1 << (bd - 1)) - 1);
    }
  }
  *s_scale = scale;
  *s_offset = offset;
}

// LIC = true: the CU tries local illumination compensation (cu.GetUseLic()).  The
// other list's prediction is then the compensated one (MotionCompensation ->
// MotionCompRefList with post_filter, inter_prediction.cc:710-722 ->
// LocalIlluminationComp :1555-1575: the model from the current reconstruction
// `rec` around the CU, neighbours in nb[job]), the full-pel stage compares with
// kSadAcOnly[Fast] and the sub-pel stage with kSatdAcOnly (GetFullpelMetric /
// GetSubpelMetric, inter_search.cc:1059-1076) - both on the int16 target; the
// candidates' own predictions stay plain (GetSubpelDist: post_filter = false).
template <int MS, bool LIC = false>
__device__ __forceinline__ void
bipred_search_body(const PlaneView &orig, const PlaneView &ref_other_arg,
                   const PlaneView &ref_search_arg, int bd, const xvcgpu_bi_block *jobs, int n,
                   xvcgpu_me_result *out, int max_launched, const PlaneView &rec,
                   const xvcgpu_mc_lic_block *nb, const RefTable *refs = nullptr,
                   const uint8_t *slots = nullptr) {
  constexpr int NW = BI_WAVES(MS);
  __shared__ BiShared<MS> s;
  __shared__ int s_scale, s_offset;
  const int ji = xcd_job_index(blockIdx.x, n);
  if (ji < 0) return;
  // (the *_refs form: slots[2 * job] = the searched picture, [2 * job + 1] = the other)
  int slot_s = 0, slot_o = 0;
  if (slots) {
    slot_s = __builtin_amdgcn_readfirstlane((int)slots[2 * ji]);
    slot_o = __builtin_amdgcn_readfirstlane((int)slots[2 * ji + 1]);
    if (slot_s >= refs->n || slot_o >= refs->n) return;
  }
  const PlaneView ref_other = slots ? refs->pic[slot_o].c[0] : ref_other_arg;
  const PlaneView ref_search = slots ? refs->pic[slot_s].c[0] : ref_search_arg;
  const xvcgpu_bi_block job = jobs[ji];
  const xvcgpu_me_block &b = job.blk;
  {  // block-size class of this kernel instance (LDS footprint)
    const int m = b.w > b.h ? b.w : b.h;
    const bool pow2 = (b.w & (b.w - 1)) == 0 && (b.h & (b.h - 1)) == 0;
    const bool valid = pow2 && b.w >= 4 && b.h >= 4 && m <= 64 && m <= max_launched;
    if (MS == 16 && !valid) {  // nobody takes it: the XVCGPU_ME_UNSUPPORTED record
      if (threadIdx.x == 0) {
        xvcgpu_me_result r;
        r.fullpel_x = r.fullpel_y = r.mv_x = r.mv_y = 0;
        r.fullpel_cost = r.subpel_dist = 0xffffffffu;
        out[ji] = r;
      }
      return;
    }
    if (!valid || (MS == 16 && m > 16) || (MS == 32 && (m <= 16 || m > 32)) ||
        (MS == 64 && m <= 32))
      return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = b.w, h = b.h, lw = 31 - __clz(w);
  int16_t *tmp = s.wv[wave].tmp;
  uint16_t *pred = s.wv[wave].pred;

  // prediction from the other list, then target = 2*orig - pred
  bi_mc_luma_slabs(bd, b, ref_other, job.other_mv_x, job.other_mv_y, tmp, s.wv[0].pred, wave, NW);
  __syncthreads();
  if (LIC) {
    const xvcgpu_mc_lic_block q = nb[ji];
    int cmx = job.other_mv_x, cmy = job.other_mv_y;
    d_clip_mv(b.x, b.y, ref_other.w, ref_other.h, cmx, cmy);
    wg_lic_model(bd, 0, b.x, b.y, w, h, cmx, cmy, q.neighbors, q.above_x, q.above_y, q.left_x,
                 q.left_y, ref_other.w, ref_other.h, ref_other, rec, &s_scale, &s_offset);
    __syncthreads();
    const int scale = s_scale, offset = s_offset, smax = (1 << bd) - 1;
    uint16_t *p0 = s.wv[0].pred;
    for (int i = threadIdx.x; i < w * h; i += 64 * NW)
      p0[i] = (uint16_t)d_clip3(((scale * (int)p0[i]) >> 5) + offset, 0, smax);
    __syncthreads();
  }
  {
    const uint16_t *o = orig.p + (ptrdiff_t)b.y * orig.stride + b.x;
    const uint16_t *p0 = s.wv[0].pred;
    for (int i = threadIdx.x; i < w * h; i += 64 * NW)
      s.target[i] = (int16_t)(2 * (int)o[(ptrdiff_t)(i >> lw) * orig.stride + (i & (w - 1))] -
                              (int)p0[i]);
  }
  __syncthreads();

  // FullSearch: window = DetermineMinMaxMv(mv_bootstrap, 4)
  int mnx, mny, mxx, mxy;
  d_min_max_mv(b.x, b.y, ref_search.w, ref_search.h, job.boot_mv_x, job.boot_mv_y, 4,
               mnx, mny, mxx, mxy);
  const int nx = mxx - mnx + 1, ny = mxy - mny + 1;
  const int down = b.fullpel_mv ? 2 : 0;
  const uint16_t *rcu = ref_search.p + (ptrdiff_t)b.y * ref_search.stride + b.x;
  unsigned long long best = ~0ull;
  for (int c = wave; c < nx * ny; c += NW) {
    const int my = mny + c / nx, mx = mnx + c % nx;
    const uint16_t *r = rcu + (ptrdiff_t)my * ref_search.stride + mx;
    unsigned long long dist;
    if (LIC)    // kSadAcOnlyFast / kSadAcOnly
      dist = wave_sad_ac(h > 8 ? 1 : 0, bd, w, h, s.target, w, r, ref_search.stride);
    else if (h > 8)  // kSadFast
      dist = ((unsigned long long)(long long)wave_sad(w, h / 2, 2, s.target, w, r,
                                                      ref_search.stride) * 2) >> (bd - 8);
    else
      dist = (unsigned long long)(long long)wave_sad(w, h, 1, s.target, w, r,
                                                     ref_search.stride) >> (bd - 8);
    const uint32_t bits = d_mvd_bits_fullpel(b.mvp_x, b.mvp_y, mx, my, down);
    const unsigned long long cost = dist + ((b.lambda16 * bits) >> 16);
    const unsigned long long key = (cost << 8) | (unsigned)c;
    best = key < best ? key : best;
  }
  if (lane == 0) s.key[wave] = best;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NW; k++) best = s.key[k] < best ? s.key[k] : best;
  const int fc = (int)(best & 0xff);
  const int fpx = mnx + fc % nx, fpy = mny + fc / nx;
  __syncthreads();

  int bx = fpx * 16, by = fpy * 16;
  uint32_t bdist;
  if (b.fullpel_mv) {
    if (wave == 0) {
      bi_mc_luma(bd, b, ref_search, bx, by, tmp, pred);
      wave_sync();
      const int avg = LIC ? wave_mean_diff(0, w, h, s.target, w, pred, w) : 0;
      const uint32_t d = (uint32_t)wave_satd(bd, w, h, avg, s.target, w, pred, w);
      if (lane == 0) s.dist[0] = d;
    }
    __syncthreads();
    bdist = s.dist[0];
  } else {
    unsigned long long carry = ~0ull;  // best cost so far (key form, idx 0)
    for (int pass = 0; pass < 2; pass++) {
      const int scale = pass == 0 ? 8 : 4;
      best = ~0ull;
      for (int i = pass + wave; i < 9; i += NW) {
        const int8_t *d = pass == 0 ? kSubpelOff[0][i] : kSubpelOff[1][i];
        const int mx = bx + d[0] * scale, my = by + d[1] * scale;
        bi_mc_luma(bd, b, ref_search, mx, my, tmp, pred);
        wave_sync();
        const int avg = LIC ? wave_mean_diff(0, w, h, s.target, w, pred, w) : 0;
        const uint32_t dist = (uint32_t)wave_satd(bd, w, h, avg, s.target, w, pred, w);
        wave_sync();
        const uint32_t bits = d_mvd_bits(b.mvp_x, b.mvp_y, mx, my, 0);
        const unsigned long long cost =
            (unsigned long long)dist + ((b.lambda16 * bits) >> 16);
        const unsigned long long key = (cost << 8) | (unsigned)i;
        best = key < best ? key : best;
        if (lane == 0) s.dist[pass * 9 + i] = dist;
      }
      if (lane == 0) s.key[wave] = best;
      __syncthreads();
      best = carry;
#pragma unroll
      for (int k = 0; k < NW; k++) best = s.key[k] < best ? s.key[k] : best;
      const int bi = (int)(best & 0xff);
      if (pass == 0 || bi != 0) {
        const int8_t *d = pass == 0 ? kSubpelOff[0][bi] : kSubpelOff[1][bi];
        bdist = s.dist[pass * 9 + bi];
        bx += d[0] * scale;
        by += d[1] * scale;
      }
      carry = best & ~0xffull;  // same cost, index 0: later equal costs lose
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    xvcgpu_me_result r;
    r.fullpel_x = fpx;
    r.fullpel_y = fpy;
    r.mv_x = bx;
    r.mv_y = by;
    r.fullpel_cost = 0;
    r.subpel_dist = bdist >> 1;  // inter_search.cc:660
    out[ji] = r;
  }
}

template <int MS, bool LIC = false>
__global__ void __launch_bounds__(64 * BI_WAVES(MS))
bipred_search_kernel(PlaneView orig, PlaneView ref_other, PlaneView ref_search,
                     int bd, const xvcgpu_bi_block *jobs, int n,
                     xvcgpu_me_result *out, int max_launched, PlaneView rec = PlaneView(),
                     const xvcgpu_mc_lic_block *nb = nullptr) {
  bipred_search_body<MS, LIC>(orig, ref_other, ref_search, bd, jobs, n, out, max_launched, rec, nb);
}

// The refinement steps of one CU state into several pairs of reference pictures in
// one launch (xvcgpu_bipred_search_refs): job i searches refs.pic[slots[2 i]] against
// the prediction from refs.pic[slots[2 i + 1]]; a slot beyond the table: no job.
template <int MS>
__global__ void __launch_bounds__(64 * BI_WAVES(MS))
bipred_search_refs_kernel(PlaneView orig, RefTable refs, const uint8_t *slots, int bd,
                          const xvcgpu_bi_block *jobs, int n, xvcgpu_me_result *out,
                          int max_launched) {
  bipred_search_body<MS, false>(orig, orig, orig, bd, jobs, n, out, max_launched, PlaneView(),
                                nullptr, &refs, slots);
}

// MotionCompUniPred -> int16 (14-bit, offset removed) by the workgroup
// (inter_prediction.cc:1156-1172; FilterCopyBipred :1462-1473; shift/offset
// rules inter_prediction.h:218-254).  tmp: w * (h + N - 1); dst stride w.
// Contains __syncthreads(): call uniformly.
template <bool CHROMA>
__device__ __forceinline__ void wg_interp_block_bipred(int bd, int w, int h, int fx,
                                                       int fy, const uint16_t *ref,
                                                       int rs, int16_t *tmp,
                                                       int16_t *dst) {
  constexpr int N = CHROMA ? 4 : 8;
  constexpr int BACK = N / 2 - 1;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lw = 31 - __clz(w);
  const int16_t *fh = CHROMA ? kChromaTaps[fx] : kLumaTaps[fx];
  const int16_t *fv = CHROMA ? kChromaTaps[fy] : kLumaTaps[fy];
  const int head = 14 - bd;
  const int sh1 = 6 - head, off1 = -(8192 << sh1);  // Sample -> int16
  if (fx == 0 && fy == 0) {
    for (int i = tid; i < w * h; i += nt) {
      const int16_t v = (int16_t)(ref[(ptrdiff_t)(i >> lw) * rs + (i & (w - 1))] << head);
      dst[i] = (int16_t)(v - (int16_t)8192);
    }
    return;
  }
  if (fy == 0 || fx == 0) {
    const int16_t *f = fy == 0 ? fh : fv;
    const ptrdiff_t step = fy == 0 ? 1 : rs;
    for (int i = tid; i < w * h; i += nt) {
      const uint16_t *s = ref + (ptrdiff_t)(i >> lw) * rs + (i & (w - 1)) - BACK * step;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k * step] * f[k];
      dst[i] = (int16_t)((sum + off1) >> sh1);
    }
    return;
  }
  for (int i = tid; i < w * (h + N - 1); i += nt) {
    const uint16_t *s = ref + (ptrdiff_t)((i >> lw) - BACK) * rs + (i & (w - 1)) - BACK;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < N; k++) sum += (int)s[k] * fh[k];
    tmp[i] = (int16_t)((sum + off1) >> sh1);
  }
  __syncthreads();
  for (int i = tid; i < w * h; i += nt) {  // int16 -> int16: shift 6, offset 0
    const int16_t *s = tmp + i;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < N; k++) sum += (int)s[k * w] * fv[k];
    dst[i] = (int16_t)(sum >> 6);
  }
}

// grid: n; block: 256.  ref0 / ref1 are the list-0 / list-1 pictures.
__global__ void __launch_bounds__(256)
mc_bipred_kernel(PicView ref0, PicView ref1, PicView pred,
                 const xvcgpu_mc_bi_block *blocks, int n) {
  __shared__ int16_t tmp[64 * 71];
  __shared__ int16_t p[2][64 * 64];
  const int bi = blockIdx.x;
  if (bi >= n) return;
  const xvcgpu_mc_bi_block b = blocks[bi];
  const int bd = ref0.bd;
  const int cs = b.comp ? 1 : 0, shift = 4 + cs;
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  for (int l = 0; l < 2; l++) {
    int mx = l ? b.mv1_x : b.mv0_x, my = l ? b.mv1_y : b.mv0_y;
    const PlaneView pr = l ? ref1.c[b.comp] : ref0.c[b.comp];
    d_clip_mv(b.x, b.y, ref0.c[0].w, ref0.c[0].h, mx, my);
    const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
    const uint16_t *r = pr.p + (ptrdiff_t)(cy + (my >> shift)) * pr.stride + cx + (mx >> shift);
    __syncthreads();  // tmp reuse
    if (b.comp)
      wg_interp_block_bipred<true>(bd, cw, ch, fx, fy, r, pr.stride, tmp, p[l]);
    else
      wg_interp_block_bipred<false>(bd, cw, ch, fx, fy, r, pr.stride, tmp, p[l]);
  }
  __syncthreads();
  // AddAvgBi (inter_prediction.cc:1545-1547)
  const int head = 14 - bd;
  const int sh = (head > 2 ? head : 2) + 1;
  const int off = (1 << (sh - 1)) + 2 * 8192;
  const int smax = (1 << bd) - 1;
  const PlaneView pd = pred.c[b.comp];
  uint16_t *dst = pd.p + (ptrdiff_t)cy * pd.stride + cx;
  const int lw = 31 - __clz(cw);
  for (int i = threadIdx.x; i < cw * ch; i += 256)
    dst[(ptrdiff_t)(i >> lw) * pd.stride + (i & (cw - 1))] =
        d_clip_bd(((int)p[0][i] + (int)p[1][i] + off) >> sh, smax);
}

// I3 (affine half): MotionCompAffine -> Sample (inter_prediction.cc:1044-1136).
// One workgroup per (CU, component); the sub-blocks are dealt to the 4 waves.
__device__ __forceinline__ int d_affine_subblock(int rx, int ry, int mx, int my, int size,
                                                 int scale) {
  const int dx = d_abs(mx - rx), dy = d_abs(my - ry);
  const int max_len = dx > dy ? dx : dy;
  if (!max_len) return size;
  int sb = (size >> 2) / max_len;
  sb = sb < 1 ? 1 : sb;
  while (size % sb) sb--;
  return (sb > 4 ? sb : 4) >> scale;
}

__global__ void __launch_bounds__(256)
mc_affine_kernel(PicView ref, PicView pred, const xvcgpu_mc_affine_block *blocks, int n) {
  __shared__ union {
    int16_t whole[64 * 71];  // mv[0] == mv[1]: plain MC of the block
    struct {
      int16_t tmp[16 * 71];
      uint16_t dst[16 * 64];
    } wv[4];
  } sh;
  const int bi = blockIdx.x;
  if (bi >= n) return;
  const xvcgpu_mc_affine_block b = blocks[bi];
  const int bd = ref.bd, comp = b.comp;
  const int pic_w = ref.c[0].w, pic_h = ref.c[0].h;
  int mv[3][2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    mv[i][0] = b.mv[i][0];
    mv[i][1] = b.mv[i][1];
    d_clip_mv(b.x, b.y, pic_w, pic_h, mv[i][0], mv[i][1]);
  }
  const int cs = comp ? 1 : 0, shift = 4 + cs, mask = (1 << shift) - 1;
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  const PlaneView pr = ref.c[comp], pd = pred.c[comp];
  uint16_t *out = pd.p + (ptrdiff_t)cy * pd.stride + cx;
  if (mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) {
    const uint16_t *r =
        pr.p + (ptrdiff_t)(cy + (mv[0][1] >> shift)) * pr.stride + cx + (mv[0][0] >> shift);
    if (comp)
      wg_interp_block<true>(bd, cw, ch, mv[0][0] & mask, mv[0][1] & mask, r, pr.stride,
                            sh.whole, out, pd.stride);
    else
      wg_interp_block<false>(bd, cw, ch, mv[0][0] & mask, mv[0][1] & mask, r, pr.stride,
                             sh.whole, out, pd.stride);
    return;
  }
  const int sbw = d_affine_subblock(mv[0][0], mv[0][1], mv[1][0], mv[1][1], cw, cs);
  const int sbh = d_affine_subblock(mv[0][0], mv[0][1], mv[2][0], mv[2][1], ch, cs);
  const int mv_max_x = (pic_w - b.x + 8 - 1) * 16, mv_min_x = (-64 - b.x - 8 + 1) * 16;
  const int mv_max_y = (pic_h - b.y + 8 - 1) * 16, mv_min_y = (-64 - b.y - 8 + 1) * 16;
  const int dhx = ((mv[1][0] - mv[0][0]) * 256) / cw;  // C division
  const int dhy = ((mv[1][1] - mv[0][1]) * 256) / cw;
  const int dvx = -dhy, dvy = dhx;
  const int nsx = cw / sbw, nsy = ch / sbh;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lsw = 31 - __clz(sbw);
  for (int k = wave; k < nsx * nsy; k += 4) {
    const int iy = k / nsx, ix = k - iy * nsx;
    // the reference's running sums, in closed form
    const int hor_x = mv[0][0] * 256 + dvx * sbh * iy + dhx * sbw * ix;
    const int hor_y = mv[0][1] * 256 + dvy * sbh * iy + dhy * sbw * ix;
    int mx = (hor_x + dhx * (sbw >> 1) + dvx * (sbh >> 1)) >> 8;
    int my = (hor_y + dhy * (sbw >> 1) + dvy * (sbh >> 1)) >> 8;
    mx = d_clip3(mx, mv_min_x, mv_max_x);
    my = d_clip3(my, mv_min_y, mv_max_y);
    const int sx = ix * sbw, sy = iy * sbh;
    const uint16_t *r =
        pr.p + (ptrdiff_t)(cy + sy + (my >> shift)) * pr.stride + cx + sx + (mx >> shift);
    wave_sync();  // previous sub-block of this wave copied out
    if (comp)
      wave_interp_block<true>(bd, sbw, sbh, mx & mask, my & mask, r, pr.stride,
                              sh.wv[wave].tmp, sh.wv[wave].dst);
    else
      wave_interp_block<false>(bd, sbw, sbh, mx & mask, my & mask, r, pr.stride,
                               sh.wv[wave].tmp, sh.wv[wave].dst);
    wave_sync();
    uint16_t *o = out + (ptrdiff_t)sy * pd.stride + sx;
    for (int i = lane; i < sbw * sbh; i += 64)
      o[(ptrdiff_t)(i >> lsw) * pd.stride + (i & (sbw - 1))] = sh.wv[wave].dst[i];
  }
}

// T4 building block (GetSubpelDist / EvalStartMvp / SearchMergeCandidates per
// candidate): MotionCompensationMv of the luma block into LDS, then
// SampleMetric::CompareSample(orig, pred).  One wave per candidate, two per
// workgroup.  grid: ceil(n/2); block: 128.
__device__ __forceinline__ void
mc_metric_body(const PlaneView &orig, const PlaneView &ref_arg, int bd, int strength,
               const xvcgpu_mc_metric_cand *cands, int n, uint64_t *out,
               const RefTable *refs = nullptr, const uint8_t *slots = nullptr) {
  __shared__ struct {
    int16_t tmp[64 * 71];
    uint16_t pred[64 * 64];
  } sh[2];
  const int wave = threadIdx.x >> 6;
  const int ci = blockIdx.x * 2 + wave;
  if (ci >= n) return;
  int slot = 0;
  if (slots) {
    slot = __builtin_amdgcn_readfirstlane((int)slots[ci]);   // (one candidate per wave)
    if (slot >= refs->n) return;
  }
  const PlaneView ref = slots ? refs->pic[slot].c[0] : ref_arg;
  const xvcgpu_mc_metric_cand cd = cands[ci];
  xvcgpu_me_block b;
  b.x = cd.x;
  b.y = cd.y;
  b.w = cd.w;
  b.h = cd.h;
  bi_mc_luma(bd, b, ref, cd.mv_x, cd.mv_y, sh[wave].tmp, sh[wave].pred);
  wave_sync();
  const uint16_t *o = orig.p + (ptrdiff_t)cd.y * orig.stride + cd.x;
  const uint64_t dist = wave_compare(cd.metric, bd, cd.qp, strength, cd.w, cd.h, o,
                                     orig.stride, sh[wave].pred, cd.w);
  if ((threadIdx.x & 63) == 0) out[ci] = dist;
}

__global__ void __launch_bounds__(128)
mc_metric_kernel(PlaneView orig, PlaneView ref, int bd, int strength,
                 const xvcgpu_mc_metric_cand *cands, int n, uint64_t *out) {
  mc_metric_body(orig, ref, bd, strength, cands, n, out);
}

// candidate i predicted from refs.pic[slots[i]] (xvcgpu_mc_metric_batch_refs)
__global__ void __launch_bounds__(128)
mc_metric_refs_kernel(PlaneView orig, RefTable refs, const uint8_t *slots, int bd, int strength,
                      const xvcgpu_mc_metric_cand *cands, int n, uint64_t *out) {
  mc_metric_body(orig, orig, bd, strength, cands, n, out, &refs, slots);
}

#endif  // XVCGPU_K_BIPRED_H_
