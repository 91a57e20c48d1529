// k_interp.h -- I1: InterPrediction::MotionCompUniPred -> Sample
// (inter_prediction.cc:1138-1154, FilterLuma/FilterChroma :1387-1448, C
// kernels :1207-1385, shift/offset rules inter_prediction.h:218-254) as a
// workgroup-cooperative device function.
//
// All threads of the workgroup call wg_interp_block() with uniform arguments.
// The horizontal pass reads the reference window from global memory (L1/L2
// resident: neighbouring CUs and the 17 sub-pel candidates of one CU hit the
// same lines), writes the 14-bit intermediate to LDS; the vertical pass reads
// LDS columns.  Output goes to `dst` (LDS or global), row stride `ds`.
#ifndef XVCGPU_K_INTERP_H_
#define XVCGPU_K_INTERP_H_

#include "dev_common.h"
#include "dev_tables.h"

// tmp must hold w * (h + 7) int16.  Contains __syncthreads(): call uniformly.
template <bool CHROMA>
__device__ __forceinline__ void wg_interp_block(int bd, int w, int h, int fx,
                                                int fy, const uint16_t *ref,
                                                int rs, int16_t *tmp,
                                                uint16_t *dst, int ds) {
  constexpr int N = CHROMA ? 4 : 8;
  constexpr int BACK = N / 2 - 1;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int smax = (1 << bd) - 1;
  const int lw = 31 - __clz(w);
  const int16_t *fh = CHROMA ? kChromaTaps[fx] : kLumaTaps[fx];
  const int16_t *fv = CHROMA ? kChromaTaps[fy] : kLumaTaps[fy];
  if (fx == 0 && fy == 0) {  // CopyFrom
    for (int i = tid; i < w * h; i += nt) {
      const int y = i >> lw, x = i & (w - 1);
      dst[y * ds + x] = ref[(ptrdiff_t)y * rs + x];
    }
    return;
  }
  if (fy == 0) {  // FilterHorSampleSample
    for (int i = tid; i < w * h; i += nt) {
      const int y = i >> lw, x = i & (w - 1);
      const uint16_t *s = ref + (ptrdiff_t)y * rs + x - BACK;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k] * fh[k];
      dst[y * ds + x] = d_clip_bd((sum + 32) >> 6, smax);
    }
    return;
  }
  if (fx == 0) {  // FilterVerSampleSample (narrows to int16 before the clip)
    for (int i = tid; i < w * h; i += nt) {
      const int y = i >> lw, x = i & (w - 1);
      const uint16_t *s = ref + (ptrdiff_t)(y - BACK) * rs + x;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[(ptrdiff_t)k * rs] * fv[k];
      dst[y * ds + x] = d_clip_bd((int16_t)((sum + 32) >> 6), smax);
    }
    return;
  }
  // FilterHorSampleShort over h + N - 1 rows, then FilterVerShortSample
  {
    const int shift = 6 - (14 - bd);
    const int offset = -(8192 << shift);
    const int rows = h + N - 1;
    for (int i = tid; i < w * rows; i += nt) {
      const int y = i >> lw, x = i & (w - 1);
      const uint16_t *s = ref + (ptrdiff_t)(y - BACK) * rs + x - BACK;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k] * fh[k];
      tmp[i] = (int16_t)((sum + offset) >> shift);
    }
  }
  __syncthreads();
  {
    const int shift = 6 + (14 - bd);
    const int offset = (8192 << 6) + (1 << (shift - 1));
    for (int i = tid; i < w * h; i += nt) {
      const int y = i >> lw, x = i & (w - 1);
      const int16_t *s = tmp + y * w + x;
      int sum = 0;
#pragma unroll
      for (int k = 0; k < N; k++) sum += (int)s[k * w] * fv[k];
      dst[y * ds + x] = d_clip_bd((int16_t)((sum + offset) >> shift), smax);
    }
  }
}

#endif  // XVCGPU_K_INTERP_H_
