// dev_common.h -- device helpers (gfx950, wave64).
#ifndef XVCGPU_DEV_COMMON_H_
#define XVCGPU_DEV_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define XVC_WAVE 64

__device__ __forceinline__ int d_clip3(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}
__device__ __forceinline__ int d_abs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int d_log2_size(int size) {  // util::SizeToLog2
  int l = 1;
  while ((1 << l) < size) l++;
  return l;
}
__device__ __forceinline__ uint16_t d_clip_bd(int v, int smax) {
  return (uint16_t)(v < 0 ? 0 : (v > smax ? smax : v));
}

// Sum over a group of `G` consecutive lanes (G power of two <= 64); every lane
// of the group receives the total.
template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int s = 1; s < G; s <<= 1) v += __shfl_xor(v, s, XVC_WAVE);
  return v;
}

// Arguments of one kernel for up to XVC_MULTI_MAX pictures, passed by value: a
// "multi" kernel is the single-picture kernel's body run with the arguments of
// picture blockIdx.y (xvcgpu_frame_pass_multi: kernels of the same kind run well
// beside each other, kernels of different kinds do not - profiles/ARCHIVE_r01_r04_design_measured.md section 6b).
#define XVC_MULTI_MAX 4
template <typename A>
struct MultiArgs {
  A a[XVC_MULTI_MAX];
};

// ClipMv, inter_prediction.cc:769-782 (1/16-pel units).
__device__ __forceinline__ void d_clip_mv(int pos_x, int pos_y, int pic_w,
                                          int pic_h, int &mx, int &my) {
  const int min_x = -((64 + 8 + pos_x - 1) << 4);
  const int min_y = -((64 + 8 + pos_y - 1) << 4);
  const int max_x = (pic_w + 8 - pos_x - 1) << 4;
  const int max_y = (pic_h + 8 - pos_y - 1) << 4;
  mx = d_clip3(mx, min_x, max_x);
  my = d_clip3(my, min_y, max_y);
}

// GetNumExpGolombBits, inter_search.cc:1179-1188 (closed form: the loop adds
// 2 per halving until 1 => 2*floor(log2(u)) + 1).
__device__ __forceinline__ uint32_t d_eg_bits(int mvd) {
  // u = mvd <= 0 ? (-mvd << 1) + 1 : mvd << 1; for mvd > 0, 2 mvd and 2 mvd + 1 have the
  // same leading bit, so (|mvd| << 1) | 1 serves both signs
  const uint32_t u = ((uint32_t)(mvd < 0 ? -mvd : mvd) << 1) | 1u;
  return 63u - 2u * (uint32_t)__clz((int)u);
}
// GetMvdBitsFullpel, inter_search.cc:1166-1177
__device__ __forceinline__ uint32_t d_mvd_bits_fullpel(int mvp_x, int mvp_y,
                                                       int fx, int fy,
                                                       int down) {
  down += 2;
  return d_eg_bits(((fx * 16) - mvp_x) >> down) +
         d_eg_bits(((fy * 16) - mvp_y) >> down);
}
// GetMvdBits, inter_search.cc:1150-1159
__device__ __forceinline__ uint32_t d_mvd_bits(int mvp_x, int mvp_y, int mx,
                                               int my, int down) {
  return d_eg_bits((mx - mvp_x) >> (2 + down)) +
         d_eg_bits((my - mvp_y) >> (2 + down));
}

// XCD-aware job index (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroup
// b is observed to run on XCD b % 8).  Jobs are laid out in picture raster
// order, so giving XCD k the k-th contiguous eighth of the job list keeps each
// L2's working set to one horizontal band of the pictures instead of all of
// them.  Launch ceil(n/8)*8 workgroups; returns -1 for the padding ones.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_job_index(int block, int n) {
  const int chunk = (n + 7) >> 3;
  const int job = (block & 7) * chunk + (block >> 3);
  return ((block >> 3) < chunk && job < n) ? job : -1;
}

#endif  // XVCGPU_DEV_COMMON_H_
