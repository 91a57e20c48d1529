// k_misc.h -- I1 batched motion compensation and the picture SSD walk.
#ifndef XVCGPU_K_MISC_H_
#define XVCGPU_K_MISC_H_

#include "dev_common.h"
#include "k_interp.h"
#include "k_metric.h"
#include "xvcgpu_internal.h"

// One workgroup = InterPrediction::MotionCompensationMv for one component of
// one uni-pred CU (inter_prediction.cc:740-758, GetFullpelRef :1174-1205,
// 4:2:0).  grid: n; block: 256.
__global__ void __launch_bounds__(256)
mc_batch_kernel(PicView ref, PicView pred, const xvcgpu_mc_block *blocks, int n) {
  __shared__ int16_t tmp[64 * 71];
  const int bi = blockIdx.x;
  if (bi >= n) return;
  const xvcgpu_mc_block b = blocks[bi];
  int mx = b.mv_x, my = b.mv_y;
  d_clip_mv(b.x, b.y, ref.c[0].w, ref.c[0].h, mx, my);
  const int cs = b.comp ? 1 : 0;
  const int shift = 4 + cs;
  const int pel_x = mx >> shift, pel_y = my >> shift;
  // chroma: frac = (mv & mask) << (1 - size_shift) = << 0 for 4:2:0
  const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
  const PlaneView pr = ref.c[b.comp], pd = pred.c[b.comp];
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  const uint16_t *r = pr.p + (ptrdiff_t)(cy + pel_y) * pr.stride + cx + pel_x;
  uint16_t *dst = pd.p + (ptrdiff_t)cy * pd.stride + cx;
  if (b.comp)
    wg_interp_block<true>(ref.bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
  else
    wg_interp_block<false>(ref.bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
}

// MotionCompensationMv of a CU with local illumination compensation
// (inter_prediction.cc:740-758 -> LocalIlluminationComp :1555-1575 ->
// DeriveLicParams :1577-1663).  One workgroup per job: the ordinary prediction
// into `pred`, then the linear model from the row above / column left of the
// block (current reconstruction `rec` against the reference displaced by the
// rounded full-pel vector, which - as in the reference - passes through ClipMv
// of the neighbouring CU unchanged in units), applied in place.
// grid: n; block: 256.
__global__ void __launch_bounds__(256)
mc_lic_kernel(PicView ref, PicView rec, PicView pred, const xvcgpu_mc_lic_block *blocks,
              int n) {
  __shared__ int16_t tmp[64 * 71];
  __shared__ int s_scale, s_offset;
  const int bi = blockIdx.x;
  if (bi >= n) return;
  const xvcgpu_mc_lic_block b = blocks[bi];
  const int pic_w = ref.c[0].w, pic_h = ref.c[0].h, bd = ref.bd;
  int mx = b.mv_x, my = b.mv_y;
  d_clip_mv(b.x, b.y, pic_w, pic_h, mx, my);
  const int cs = b.comp ? 1 : 0;
  const int shift = 4 + cs;
  const int pel_x = mx >> shift, pel_y = my >> shift;
  const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
  const PlaneView pr = ref.c[b.comp], pd = pred.c[b.comp], pc = rec.c[b.comp];
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  uint16_t *dst = pd.p + (ptrdiff_t)cy * pd.stride + cx;
  {
    const uint16_t *r = pr.p + (ptrdiff_t)(cy + pel_y) * pr.stride + cx + pel_x;
    if (b.comp)
      wg_interp_block<true>(bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
    else
      wg_interp_block<false>(bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
  }
  // model: the first wave sums the (<= 64) neighbour pairs, lane 0 solves
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool has_above = b.neighbors & XVC_LIC_HAS_ABOVE, has_left = b.neighbors & XVC_LIC_HAS_LEFT;
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
        d_clip_mv(b.above_x, b.above_y, pic_w, pic_h, vx, vy);
        a = rb[(ptrdiff_t)(vy - 1) * pr.stride + vx + i * dx];
        d = sb[-(ptrdiff_t)pc.stride + i * dx];
      } else {
        int vx = full_x, vy = full_y;
        d_clip_mv(b.left_x, b.left_y, pic_w, pic_h, vx, vy);
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
    if (lane == 0) {
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
      s_scale = scale;
      s_offset = offset;
    }
  }
  __syncthreads();  // also orders the prediction stores before the reads below
  const int scale = s_scale, offset = s_offset, smax = (1 << bd) - 1;
  const int lw = 31 - __clz(cw);
  for (int i = threadIdx.x; i < cw * ch; i += 256) {
    uint16_t *p = dst + (ptrdiff_t)(i >> lw) * pd.stride + (i & (cw - 1));
    *p = (uint16_t)d_clip3(((scale * (int)*p) >> 5) + offset, 0, smax);
  }
}

// Same, with the MV taken from the motion search result of the CU: one
// workgroup per (CU, component) = InterPrediction::MotionCompensation for a
// uni-pred CU (inter_prediction.cc:710-722).  grid: (n, 3); block: 256.
__global__ void __launch_bounds__(256)
mc_from_me_kernel(PicView ref, PicView pred, const xvcgpu_me_block *blocks,
                  const xvcgpu_me_result *results, int n) {
  __shared__ int16_t tmp[64 * 71];
  const int bi = blockIdx.x, comp = blockIdx.y;
  if (bi >= n) return;
  const xvcgpu_me_block b = blocks[bi];
  int mx = results[bi].mv_x, my = results[bi].mv_y;
  d_clip_mv(b.x, b.y, ref.c[0].w, ref.c[0].h, mx, my);
  const int cs = comp ? 1 : 0;
  const int shift = 4 + cs;
  const int pel_x = mx >> shift, pel_y = my >> shift;
  const int fx = mx & ((1 << shift) - 1), fy = my & ((1 << shift) - 1);
  const PlaneView pr = ref.c[comp], pd = pred.c[comp];
  const int cx = b.x >> cs, cy = b.y >> cs, cw = b.w >> cs, ch = b.h >> cs;
  const uint16_t *r = pr.p + (ptrdiff_t)(cy + pel_y) * pr.stride + cx + pel_x;
  uint16_t *dst = pd.p + (ptrdiff_t)cy * pd.stride + cx;
  if (comp)
    wg_interp_block<true>(ref.bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
  else
    wg_interp_block<false>(ref.bd, cw, ch, fx, fy, r, pr.stride, tmp, dst, pd.stride);
}

// Host-driver glue kept on the device so pictures and decisions never leave
// HBM: fills the deblocking metadata of uni-pred inter CUs from the motion
// search results and the luma cbf of the residual pipeline (what the
// reference's CuEncoder writes into CodingUnit, cu_encoder.cc:543-577).
__device__ __forceinline__ void cu_info_from_me_kernel_body(const xvcgpu_me_block *blocks, const xvcgpu_me_result *results, const int32_t *nnz, const int32_t *luma_tx_index, int n, int qp_y, int qp_c, int ref_poc, xvcgpu_cu_info *cus) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const xvcgpu_me_block b = blocks[i];
  xvcgpu_cu_info c;
  c.x = (uint16_t)b.x;
  c.y = (uint16_t)b.y;
  c.w = b.w;
  c.h = b.h;
  c.intra = 0;
  c.cbf_luma = nnz[luma_tx_index ? luma_tx_index[i] : i] != 0;
  c.qp_y = (int8_t)qp_y;
  c.qp_c = (int8_t)qp_c;
  c.ref_idx0 = 0;
  c.reserved = 0;
  c.ref_poc[0] = ref_poc;
  c.ref_poc[1] = -1;
  for (int k = 0; k < 4; k++) {
    c.mv[0][k][0] = results[i].mv_x;
    c.mv[0][k][1] = results[i].mv_y;
    c.mv[1][k][0] = 0;
    c.mv[1][k][1] = 0;
  }
  cus[i] = c;
}

__global__ void
cu_info_from_me_kernel(const xvcgpu_me_block *blocks, const xvcgpu_me_result *results, const int32_t *nnz, const int32_t *luma_tx_index, int n, int qp_y, int qp_c, int ref_poc, xvcgpu_cu_info *cus) {
  cu_info_from_me_kernel_body(blocks, results, nnz, luma_tx_index, n, qp_y, qp_c, ref_poc, cus);
}

// SampleMetric::ComparePicture / ComputePsnr block walk (sample_metric.cc:
// 37-155).  The reference visits full 64x64 blocks only while
// x < width-64 / y < height-64 (strict) and the remainder in steps of the
// lowest set bit of the dimension, starting at (dim & ~63) - so when a
// dimension is a multiple of 64 its last block column/row is never visited.
// Each visited block is one Compare(): SSD >> 2*(bd-8), summed.
// One workgroup per visited block (the per-block `>> shift` needs the whole
// block's sum): thread = 8 samples (one 16-byte load from each picture),
// 256 / (bw / 8) rows per sweep.  grid: items; block 256.
__global__ void __launch_bounds__(256)
picture_ssd_kernel(PlaneView a, PlaneView b, int shift, int y_begin, int y_end,
                   unsigned long long *part_out) {
  __shared__ unsigned long long part[4];
  const int w = a.w, h = a.h;
  const int mbx = w & ~(w - 1), mby = h & ~(h - 1);
  const int nfx = w > 64 ? (w - 64 + 63) / 64 : 0;
  const int nrx = (w - (w & ~63)) / mbx;
  const int nfy = h > 64 ? (h - 64 + 63) / 64 : 0;
  const int ncx = nfx + nrx;
  const int item = blockIdx.x;
  const int ix = item % ncx, iy = item / ncx;
  int x, y, bw, bh;
  if (ix < nfx) { x = ix * 64; bw = 64; } else { x = (w & ~63) + (ix - nfx) * mbx; bw = mbx; }
  if (iy < nfy) { y = iy * 64; bh = 64; } else { y = (h & ~63) + (iy - nfy) * mby; bh = mby; }
  if (y < y_begin || y >= y_end) {  // another shard's block (uniform per workgroup)
    if (threadIdx.x == 0) part_out[2 * item] = part_out[2 * item + 1] = 0;
    return;
  }
  const uint16_t *pa = a.p + (ptrdiff_t)y * a.stride + x;
  const uint16_t *pb = b.p + (ptrdiff_t)y * b.stride + x;
  uint32_t acc = 0;  // <= 16 squares of 12-bit differences per thread
  if (bw >= 8) {     // picture widths are multiples of 8
    const int cpr = bw >> 3, lc = 31 - __clz(cpr);
    const int c8 = (threadIdx.x & (cpr - 1)) << 3;
    for (int r = threadIdx.x >> lc; r < bh; r += 256 >> lc) {
      const uint4 va = *reinterpret_cast<const uint4 *>(pa + (ptrdiff_t)r * a.stride + c8);
      const uint4 vb = *reinterpret_cast<const uint4 *>(pb + (ptrdiff_t)r * b.stride + c8);
      const uint32_t ua[4] = {va.x, va.y, va.z, va.w}, ub[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int d0 = (int)(ua[k] & 0xffff) - (int)(ub[k] & 0xffff);
        const int d1 = (int)(ua[k] >> 16) - (int)(ub[k] >> 16);
        acc += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
      }
    }
  } else {
    for (int i = threadIdx.x; i < bw * bh; i += 256) {
      const int r = i / bw, c = i - r * bw;
      const int d = (int)pa[(ptrdiff_t)r * a.stride + c] - (int)pb[(ptrdiff_t)r * b.stride + c];
      acc += (uint32_t)(d * d);
    }
  }
  const unsigned long long ws = group_sum<64>((unsigned long long)acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ws;
  __syncthreads();
  // per-block results go to a scratch array (no same-address atomics, no
  // zero-fill launch); picture_ssd_sum_kernel folds them
  if (threadIdx.x == 0) {
    part_out[2 * item] = (part[0] + part[1] + part[2] + part[3]) >> shift;
    part_out[2 * item + 1] = (unsigned long long)bw * bh;
  }
}

// grid: 1; block: 256.  out[0] = sum of block SSDs, out[1] = samples visited.
__device__ __forceinline__ void picture_ssd_sum_kernel_body(const unsigned long long *part_in, int items, unsigned long long *out) {
  __shared__ unsigned long long red[2][4];
  unsigned long long s0 = 0, s1 = 0;
  // batches of four independent loads per thread (one round trip per batch)
  for (int base = threadIdx.x; base < items; base += 4 * 256) {
    ulonglong2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + 256 * u;
      v[u] = i < items ? reinterpret_cast<const ulonglong2 *>(part_in)[i]
                       : make_ulonglong2(0ull, 0ull);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      s0 += v[u].x;
      s1 += v[u].y;
    }
  }
  s0 = group_sum<64>(s0);
  s1 = group_sum<64>(s1);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void __launch_bounds__(256)
picture_ssd_sum_kernel(const unsigned long long *part_in, int items, unsigned long long *out) {
  picture_ssd_sum_kernel_body(part_in, items, out);
}

// grid: (32, n segments); block 256: segment blockIdx.y, 16 bytes per thread and
// step when source, destination and length allow it, else bytes.
__global__ void __launch_bounds__(256)
copy_segments_kernel(const xvcgpu_copy_segment *segs, int n) {
  const xvcgpu_copy_segment sg = segs[blockIdx.y];
  const unsigned long long tid = blockIdx.x * 256ull + threadIdx.x, nthr = gridDim.x * 256ull;
  const uintptr_t a = reinterpret_cast<uintptr_t>(sg.src), b = reinterpret_cast<uintptr_t>(sg.dst);
  if (((a | b | sg.bytes) & 15) == 0) {
    const uint4 *s4 = static_cast<const uint4 *>(sg.src);
    uint4 *d4 = static_cast<uint4 *>(sg.dst);
    for (unsigned long long i = tid; i < sg.bytes / 16; i += nthr) d4[i] = s4[i];
  } else {
    const uint8_t *s1 = static_cast<const uint8_t *>(sg.src);
    uint8_t *d1 = static_cast<uint8_t *>(sg.dst);
    for (unsigned long long i = tid; i < sg.bytes; i += nthr) d1[i] = s1[i];
  }
}

// xvcgpu_copy_blocks: one wave per block, rows of up to 64 samples.
// grid: (n + 3) / 4; block: 256.
__global__ void __launch_bounds__(256)
copy_blocks_kernel(PicView src, PicView dst, const xvcgpu_copy_block *blocks, int n) {
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (bi >= n) return;
  const xvcgpu_copy_block b = blocks[bi];
  const PlaneView ps = src.c[b.comp], pd = dst.c[b.comp];
  const uint16_t *s = ps.p + (ptrdiff_t)b.sy * ps.stride + b.sx;
  uint16_t *d = pd.p + (ptrdiff_t)b.dy * pd.stride + b.dx;
  const int lw = 31 - __clz((int)b.w), total = b.w * b.h;
  if ((b.w & (b.w - 1)) == 0) {
    for (int i = lane; i < total; i += 64)
      d[(ptrdiff_t)(i >> lw) * pd.stride + (i & (b.w - 1))] =
          s[(ptrdiff_t)(i >> lw) * ps.stride + (i & (b.w - 1))];
  } else {
    for (int y = 0; y < b.h; y++)
      for (int x = lane; x < b.w; x += 64)
        d[(ptrdiff_t)y * pd.stride + x] = s[(ptrdiff_t)y * ps.stride + x];
  }
}

#endif  // XVCGPU_K_MISC_H_
