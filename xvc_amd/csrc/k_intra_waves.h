// k_intra_waves.h -- N1: an intra picture's reconstruction in ONE launch.
//
// The decoder reconstructs an intra picture CU by CU (CuDecoder::DecompressCu ->
// IntraPrediction::Predict + InverseTransform + AddClip, cu_decoder.cc:100-166):
// a block's prediction reads the reconstruction of its neighbours, so the
// (CU, component) jobs fall into dependency waves - 387 for a 1080p picture of
// the reference stream, a few dozen jobs each.  Launched wave by wave that is
// three dependent launches per wave (prediction, the two inverse-transform
// kernels): 1166 launches, 17 ms, nearly all of it launch latency.
//
// Here the whole picture is one cooperative launch (G workgroups resident
// together).  The jobs are sorted by wave; workgroup g takes jobs g, g + G, ...
// in that order: the block's residual (inverse transform, in LDS), then - once
// its neighbours are there - the prediction (in LDS) and the add into `rec` (job
// k of the intra list and job k of the transform list are the same block:
// xvc_gpu::PictureDecoder::Plan appends both per unit).  A job of wave w may read
// its neighbours when the counter of wave w - 1 has reached that wave's size, and
// adds itself to the counter of wave w when its block is in memory - one atomic
// per JOB on a per-wave address.  (A grid barrier
// per wave is one atomic per WORKGROUP on one address: with 512 workgroups that
// alone took 60 us per wave, 27 ms per picture.)  No deadlock: a workgroup takes
// its jobs in wave order and all workgroups are resident, so the jobs of wave
// w - 1 are always held by workgroups that are running or only wait for earlier
// waves.
#ifndef XVCGPU_K_INTRA_WAVES_H_
#define XVCGPU_K_INTRA_WAVES_H_

#include "k_intra.h"
#include "k_tx.h"

// grid: any number of workgroups that are resident together (cooperative
// launch); block: 256 (= TX_THREADS).  wave_first[w] .. wave_first[w + 1]: the
// jobs of wave w in both lists; done[n_waves]: zero on entry.
//
// What does not depend on the neighbours runs BEFORE the job waits for them: the
// block's descriptor, levels, matrices, dequantisation and inverse transform
// (the residual stays in LDS).  Behind the wait are only the reference samples,
// the prediction and the add - the part of a job that really is on the picture's
// critical path of 401 dependent steps.
__global__ void __launch_bounds__(256)
intra_waves_kernel(PicView rec, PicView pred, const xvcgpu_intra_block *jobs,
                   const xvcgpu_tx_block *blocks, const int32_t *wave_first, int n_waves,
                   int16_t *levels, const uint32_t *level_off, int32_t *nnz,
                   const int16_t *tx_tables, TxTableLayout lay, int *done) {
  __shared__ __attribute__((aligned(16))) TxShared tx;
  __shared__ IntraPredShared ip;
  __shared__ uint16_t pblk[64 * 64];
  const int n = wave_first[n_waves];
  int w = 0;
  for (int j = (int)blockIdx.x; j < n; j += (int)gridDim.x) {
    while (j >= wave_first[w + 1]) w++;
    const xvcgpu_intra_block b = jobs[j];
    // the residual of the block (0: none), in tx.a
    const int has_resi =
        residual_job<TX_MODE_INV, 4>(tx, j, pred, pred, rec, blocks, levels, level_off, nnz,
                                     tx_tables, lay, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     0, /*defer_add=*/true);
    if (w > 0) {
      // relaxed polls, then ONE agent-scope acquire by one lane: it invalidates this
      // CU's L1 for the whole workgroup (a fence per thread is 3-4 us each)
      if (threadIdx.x == 0) {
        const int need = wave_first[w] - wave_first[w - 1];
        while (__hip_atomic_load(&done[w - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need)
          __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    const PlaneView pr = rec.c[b.comp];
    const bool is_luma = b.comp == 0;
    // the prediction stays in LDS (pblk, row stride = the block's width)
    if (b.mode == XVC_INTRA_MODE_LM_CHROMA) {
      if (!is_luma && b.w <= 32 && b.h <= 32)
        intra_lm_chroma(ip.lm, b, rec.c[0], pr, rec.bd, pblk, b.w);
    } else {
      intra_build_refs<true>(ip.refs, b, pr.p + (ptrdiff_t)b.y * pr.stride + b.x, pr.stride,
                             rec.bd, is_luma, threadIdx.x, 256);
      intra_predict<true>(ip.refs, ip.line, rec.bd, is_luma, b.mode, b.w, b.h, pblk, b.w,
                          threadIdx.x, 256);
    }
    __syncthreads();
    // SampleBuffer::AddClip (sample_buffer.h:72-87) / CopyFrom for a block without levels
    {
      const int bw = b.w, bh = b.h, lw = 31 - __clz(bw), smax = (1 << rec.bd) - 1;
      for (int i = threadIdx.x; i < bw * bh; i += 256) {
        const int y = i >> lw, x = i & (bw - 1);
        const int r = has_resi ? (int)tx.a[y * TX_S + x] : 0;
        pr.p[(ptrdiff_t)(b.y + y) * pr.stride + b.x + x] =
            (uint16_t)d_clip3((int)pblk[y * bw + x] + r, 0, smax);
      }
    }
    // the block is in memory before the wave's counter says so.  A workgroup
    // barrier does not wait for the OTHER waves' outstanding stores: every wave
    // drains its own (vmcnt counts per wave) in front of the barrier; behind it one
    // lane writes the L2's dirty lines back for all of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&done[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#endif  // XVCGPU_K_INTRA_WAVES_H_
