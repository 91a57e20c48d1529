// k_cs_engine.h -- the steps of MANY pictures' CU-state chains in one launch.
//
// A CU state of the RD search (CuEncoder::CompressCu, cu_encoder.cc:123-273) is a string
// of dependent kernels of one CU's size: 4 - 30 us each on a few CUs, > 95 % of the chip
// idle.  The reference scales by coding N pictures at once (thread_encoder.cc:99-159);
// on the device N independent chains on N streams stop adding up at about four (the
// command processors run about four dependent-kernel streams of a process side by side,
// DESIGN section 6).  What does scale is more work per launch: the engine of
// xvc_amd/host/xvc_cu_state.cc takes the next step of every chain, groups the steps by
// kind and issues ONE launch per kind - grid y = the chain's step ("segment"), every
// segment with its own job arrays and its own pictures.  The kernels are the bodies the
// single-chain entry points run (same code, same results): a segment is exactly the
// argument list of one such call, the pictures come from the chain's device-resident
// environment.
#ifndef XVCGPU_K_CS_ENGINE_H_
#define XVCGPU_K_CS_ENGINE_H_

#include "k_affine_me.h"
#include "k_bipred.h"
#include "k_cu_state.h"
#include "k_inter_pred.h"
#include "k_me2.h"
#include "k_tx2.h"

// A chain's pictures, in device memory (xvcgpu_cs_env_create)
struct CsEnvDev {
  PicView orig;
  RefTable refs;
  PicView s_orig, s_pred, s_rec;
  int16_t *levels;
  xvcgpu_cs_result *results;
  int pic_w, pic_h;
};

// (88 bytes a segment: a launch's segments lie in page-locked host memory the kernels read -
// the context's ring, xvcgpu_cs_segs_launch - since more than 44 do not fit the arguments)
struct CsSegDev {
  int n, i0, r0, r1;
  const void *p[8];
  const CsEnvDev *env;
};

template <typename T>
__device__ __forceinline__ T *cs_ptr(const CsSegDev &g, int k) {
  return static_cast<T *>(const_cast<void *>(g.p[k]));
}

// xvcgpu_mc_metric_batch_refs.  grid: (max ceil(n / 2), segments); block: 128
__global__ void __launch_bounds__(128) cs_seg_mc_metric_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  mc_metric_body(e.orig.c[0], e.orig.c[0], e.orig.bd, 16, cs_ptr<const xvcgpu_mc_metric_cand>(g, 0),
                 g.n, cs_ptr<uint64_t>(g, 1), &e.refs, cs_ptr<const uint8_t>(g, 2));
}

// the folds.  grid: (max n, segments); block: 64
__global__ void cs_seg_start_fold_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  cs_start_fold_body(cs_ptr<const xvcgpu_cs_pass>(g, 0), g.i0, g.n, cs_ptr<const uint64_t>(g, 1),
                     cs_ptr<xvcgpu_me_block>(g, 2), cs_ptr<const xvcgpu_me_result>(g, 3),
                     cs_ptr<xvcgpu_affine_me_block>(g, 4), e.results, e.pic_w, e.pic_h);
}
__global__ void cs_seg_uni_fold_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  cs_uni_fold_body(cs_ptr<const xvcgpu_cs_pass>(g, 0), g.i0, g.n,
                   cs_ptr<const xvcgpu_me_result>(g, 1),
                   cs_ptr<const xvcgpu_affine_me_result>(g, 2), e.results,
                   cs_ptr<xvcgpu_bi_block>(g, 3), cs_ptr<xvcgpu_affine_me_block>(g, 4));
}
__global__ void cs_seg_bi_fold_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  cs_bi_fold_body(cs_ptr<const xvcgpu_cs_pass>(g, 0), g.i0, g.n,
                  cs_ptr<const xvcgpu_me_result>(g, 1),
                  cs_ptr<const xvcgpu_affine_me_result>(g, 2), e.results,
                  cs_ptr<xvcgpu_inter_block>(g, 3));
}
// grid: (max ceil(n / 64), segments); block: 64
__global__ void cs_seg_merge_fold_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  cs_merge_fold_body(cs_ptr<const xvcgpu_cs_merge>(g, 0), g.i0, g.n, cs_ptr<const uint64_t>(g, 1),
                     cs_ptr<const xvcgpu_inter_block>(g, 2), cs_ptr<xvcgpu_cs_merge_result>(g, 3),
                     cs_ptr<xvcgpu_inter_block>(g, 4));
}

// xvcgpu_me_search_refs (the context's rotation tables: placement only, shared by the segments).
// grid: (me2_grid(max n), segments); block: 64 * ME2_WAVES(MS)
template <int MS, int PH>
__global__ void __launch_bounds__(64 * ME2_WAVES(MS), ME2_MIN_WAVES(MS))
cs_seg_me_kernel(const CsSegDev *segs, const TzCand *tz_pattern, Me2Sched sched) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  me_search_wave_body<MS, PH, false>(e.orig, e.orig, cs_ptr<const xvcgpu_me_block>(g, 0), g.n,
                                     cs_ptr<xvcgpu_me_result>(g, 1), tz_pattern, sched, g.i0, false,
                                     &e.refs, cs_ptr<const uint8_t>(g, 2));
}
// grid: (max ceil(n / 8) * 8, segments); block: 256
__global__ void __launch_bounds__(256) cs_seg_me_team_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  me_subpel_team_body<64, 4>(e.orig, e.orig, cs_ptr<const xvcgpu_me_block>(g, 0), g.n,
                             cs_ptr<xvcgpu_me_result>(g, 1), &e.refs, cs_ptr<const uint8_t>(g, 2));
}

// xvcgpu_bipred_search_refs.  grid: (max ceil(n / 8) * 8, segments); block: 64 * BI_WAVES(MS)
template <int MS>
__global__ void __launch_bounds__(64 * BI_WAVES(MS)) cs_seg_bi_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  bipred_search_body<MS, false>(e.orig.c[0], e.orig.c[0], e.orig.c[0], e.orig.bd,
                                cs_ptr<const xvcgpu_bi_block>(g, 0), g.n,
                                cs_ptr<xvcgpu_me_result>(g, 1), g.i0, PlaneView(), nullptr, &e.refs,
                                cs_ptr<const uint8_t>(g, 2));
}

// xvcgpu_affine_me_batch_refs.  grid: (max n, segments); block: 64 * NW
template <int NW>
__global__ void __launch_bounds__(64 * NW) cs_seg_affine_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  affine_me_body<NW>(e.orig.c[0], e.orig.c[0], e.orig.c[0], e.orig.bd,
                     cs_ptr<const xvcgpu_affine_me_block>(g, 0), g.n,
                     cs_ptr<xvcgpu_affine_me_result>(g, 1), &e.refs, cs_ptr<const uint8_t>(g, 2));
}

// xvcgpu_inter_pred_batch_to into the chain's prediction scratch.  grid: (max n, segments)
__global__ void __launch_bounds__(256) cs_seg_inter_pred_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  inter_pred_body(e.refs, e.orig, e.s_pred, cs_ptr<const xvcgpu_inter_block>(g, 0), g.n,
                  cs_ptr<const xvcgpu_block_pos>(g, 1), e.pic_w, e.pic_h);
}

// xvcgpu_residual_rdoq_batch_at (r0 = the evaluation's candidates in front).
// grid: (max n + r0, segments); block: TX_THREADS
__global__ void __launch_bounds__(TX_THREADS)
cs_seg_residual_kernel(const CsSegDev *segs, const int16_t *tx_tables, const int16_t *tx_tables_t,
                       TxTableLayout lay) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  const xvcgpu_eval_cand *ec = cs_ptr<const xvcgpu_eval_cand>(g, 6);
  if ((int)blockIdx.x >= g.n + (ec ? g.r0 : 0)) return;
  residual_cu_body<TX_MODE_FULL, true>(
      e.orig, e.s_pred, e.s_rec, cs_ptr<const xvcgpu_tx_block>(g, 0), g.n, e.levels,
      cs_ptr<const uint32_t>(g, 1), cs_ptr<int32_t>(g, 2), tx_tables, tx_tables_t, lay,
      cs_ptr<const xvcgpu_rdoq_contexts>(g, 3), cs_ptr<const xvcgpu_rdoq_params>(g, 4),
      cs_ptr<const xvcgpu_block_pos>(g, 5), ec, g.r0, cs_ptr<uint64_t>(g, 7), 16);
}

// xvcgpu_eval_dist_batch against the original picture.  grid: (max ceil(n / 4), segments)
__global__ void __launch_bounds__(256) cs_seg_eval_dist_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const CsEnvDev &e = *g.env;
  eval_dist_body(e.orig, e.s_pred, e.s_rec, 16, cs_ptr<const xvcgpu_eval_cand>(g, 0), g.n,
                 cs_ptr<uint64_t>(g, 1));
}

// The chains' read-backs of a round in one launch: a segment copies n bytes from device
// memory into page-locked host memory (a copy call per read-back was 1.6 calls per CU
// state on the issuing thread).  grid: (8, segments); block: 256
__global__ void __launch_bounds__(256) cs_seg_fetch_kernel(const CsSegDev *segs) {
  const CsSegDev &g = segs[blockIdx.y];
  const uint32_t *src = cs_ptr<const uint32_t>(g, 0);
  uint32_t *dst = cs_ptr<uint32_t>(g, 1);
  const int words = g.n >> 2;
  for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < words; i += 8 * 256) dst[i] = src[i];
}

#endif  // XVCGPU_K_CS_ENGINE_H_
