// k_pad.h -- P1: YuvPicture::PadBorder (yuv_pic.cc:118-150) on device.
//
// The reference copies row 0 / row h-1 into the top / bottom border (visible
// width only) and then replicates column 0 / w-1 of EVERY padded row into the
// left / right border, so corners take the corner sample.  Equivalent order
// used here (no read-after-write hazards between workgroups of one launch):
//   kernel 1: left/right borders of the visible rows;
//   kernel 2: full padded rows 0 / h-1 copied to the rows above / below.
// HBM-bound write-only pass: 16-byte stores, rows are 256-byte aligned.
#ifndef XVCGPU_K_PAD_H_
#define XVCGPU_K_PAD_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

// grid: (h, 3); block: 64. Each wave fills both side borders of one row.
__global__ void pad_lr_kernel(PicView pic) {
  const PlaneView pl = pic.c[blockIdx.y];
  const int y = blockIdx.x;
  if (y >= pl.h) return;
  uint16_t *row = pl.p + (ptrdiff_t)y * pl.stride;
  const uint32_t l = row[0], r = row[pl.w - 1];
  const uint32_t l2 = l | (l << 16), r2 = r | (r << 16);
  const uint4 lv = make_uint4(l2, l2, l2, l2), rv = make_uint4(r2, r2, r2, r2);
  // border is a multiple of 8 samples and row - border is 16-byte aligned
  uint4 *lp = reinterpret_cast<uint4 *>(row - pl.border);
  const int nvec = pl.border / 8;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x) lp[i] = lv;
  // right border starts at row + w; w is a multiple of 4 (chroma) or 8 (luma):
  // use 8-byte stores there.
  uint2 *rp = reinterpret_cast<uint2 *>(row + pl.w);
  const int nvec2 = pl.border / 4;
  const uint2 rv2 = make_uint2(rv.x, rv.y);
  for (int i = threadIdx.x; i < nvec2; i += blockDim.x) rp[i] = rv2;
}

// grid: (2*border_luma, 3); block: 256. Copies one padded source row to one
// border row (blockIdx.x < border: above, else below).
__global__ void pad_tb_kernel(PicView pic) {
  const PlaneView pl = pic.c[blockIdx.y];
  int k = blockIdx.x;
  const bool below = k >= pl.border;
  if (below) k -= pl.border;
  if (k >= pl.border) return;  // chroma has half as many border rows
  const uint16_t *src =
      pl.p + (ptrdiff_t)(below ? pl.h - 1 : 0) * pl.stride - pl.border;
  uint16_t *dst = pl.p +
                  (ptrdiff_t)(below ? pl.h + k : -(k + 1)) * pl.stride -
                  pl.border;
  const int n = pl.w + 2 * pl.border;  // multiple of 4 samples
  const uint2 *s = reinterpret_cast<const uint2 *>(src);
  uint2 *d = reinterpret_cast<uint2 *>(dst);
  for (int i = threadIdx.x; i < n / 4; i += blockDim.x) d[i] = s[i];
}

#endif  // XVCGPU_K_PAD_H_
