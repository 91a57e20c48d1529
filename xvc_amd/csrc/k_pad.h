// k_pad.h -- P1: YuvPicture::PadBorder (yuv_pic.cc:118-150) on device.
//
// The reference copies row 0 / row h-1 into the top / bottom border (visible
// width only) and then replicates column 0 / w-1 of EVERY padded row into the
// left / right border, so corners take the corner sample.  One launch, no
// read-after-write between workgroups: a workgroup owns one padded row and
// derives everything from the visible samples of its source row
//   visible rows: left/right borders of that row;
//   border rows : corner sample | row 0 or h-1 | corner sample.
// HBM-bound write-only pass: 16-byte stores, rows are 256-byte aligned.
#ifndef XVCGPU_K_PAD_H_
#define XVCGPU_K_PAD_H_

#include "dev_common.h"
#include "xvcgpu_internal.h"

// grid: (h_luma + 2*border_luma, 3); block: 128.
__global__ void __launch_bounds__(128) pad_border_kernel(PicView pic) {
  const PlaneView pl = pic.c[blockIdx.y];
  const int k = blockIdx.x;
  if (k >= pl.h + 2 * pl.border) return;  // chroma has fewer rows
  // destination row and its source row
  int y, ys;
  if (k < pl.h) { y = ys = k; }
  else if (k < pl.h + pl.border) { y = -(k - pl.h + 1); ys = 0; }
  else { y = pl.h + (k - pl.h - pl.border); ys = pl.h - 1; }
  const uint16_t *src = pl.p + (ptrdiff_t)ys * pl.stride;
  uint16_t *row = pl.p + (ptrdiff_t)y * pl.stride;
  const uint32_t l = src[0], r = src[pl.w - 1];
  const uint32_t l2 = l | (l << 16), r2 = r | (r << 16);
  // border is a multiple of 8 samples and row - border is 16-byte aligned
  uint4 *lp = reinterpret_cast<uint4 *>(row - pl.border);
  const uint4 lv = make_uint4(l2, l2, l2, l2);
  for (int i = threadIdx.x; i < pl.border / 8; i += 128) lp[i] = lv;
  // right border starts at row + w; w is a multiple of 4 (chroma) or 8 (luma):
  // 8-byte stores there
  uint2 *rp = reinterpret_cast<uint2 *>(row + pl.w);
  const uint2 rv = make_uint2(r2, r2);
  for (int i = threadIdx.x; i < pl.border / 4; i += 128) rp[i] = rv;
  if (y != ys) {  // border row: copy the visible samples of the source row
    const uint2 *s = reinterpret_cast<const uint2 *>(src);
    uint2 *d = reinterpret_cast<uint2 *>(row);
    for (int i = threadIdx.x; i < pl.w / 4; i += 128) d[i] = s[i];
  }
}

#endif  // XVCGPU_K_PAD_H_
