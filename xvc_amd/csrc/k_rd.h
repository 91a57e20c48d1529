// k_rd.h -- C1, decision half: the strict-< folds of
// TransformEncoder::CompressAndEvalTransform (transform_encoder.cc:53-201) and the
// tail of InterSearch::CompressAndEvalCbf (inter_search.cc:316-361).  One thread
// per job: the arithmetic is a handful of IEEE double multiply-adds and integer
// compares in the reference's order; what makes it a device step is that the
// distortions it folds never leave HBM (they are the outputs of the residual
// and metric kernels) - the host only supplies the bits its entropy coder
// prices the alternatives with.
#ifndef XVCGPU_K_RD_H_
#define XVCGPU_K_RD_H_

#include "dev_common.h"

// dist + static_cast<Cost>(bits * lambda + 0.5) (transform_encoder.cc:91, :139)
__device__ __forceinline__ unsigned long long rd_cost(unsigned long long dist, unsigned bits,
                                                       double lambda) {
  return dist + (unsigned long long)((double)bits * lambda + 0.5);
}

__global__ void __launch_bounds__(256)
tx_eval_kernel(const xvcgpu_tx_eval_job *jobs, int n, const xvcgpu_tx_eval_alt *alts,
               xvcgpu_tx_eval_result *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const xvcgpu_tx_eval_job j = jobs[i];
  const unsigned long long kMax = 0xffffffffffffffffull;
  // best_cost = { max, 0, 0 }, or the previous pass' cost (:93-96)
  unsigned long long best_cost = j.prev_cost, best_reco = 0, best_resi = 0;
  int best = -2;
  bool best_cbf = (j.flags & XVC_TXE_PREV_CBF) != 0;
  for (int a = 0; a < j.n_alt; a++) {
    const xvcgpu_tx_eval_alt t = alts[j.alt_first + a];
    if (t.kind == XVC_TXE_KIND_SELECT && (j.flags & XVC_TXE_FAST_SELECT) && !best_cbf) break;
    // get_transform_cost (:64-92): an invalid alternative costs max()
    const bool invalid = t.dist_reco == XVC_TXE_DIST_INVALID;
    const unsigned long long cost = invalid ? kMax : rd_cost(t.dist_resi, t.bits, j.lambda);
    if (cost < best_cost) {
      best_cost = cost;
      best_reco = t.dist_reco;
      best_resi = invalid ? t.dist_reco : t.dist_resi;
      best = a;
      best_cbf = t.cbf != 0;
    }
    // the all-zero block, right after the default transform (:112-144); it only
    // competes when that transform kept a coefficient (cu->GetCbf(comp))
    if (t.kind == XVC_TXE_KIND_NORMAL && (j.flags & XVC_TXE_CBF_ZERO) && t.cbf && !invalid) {
      const unsigned long long zc = rd_cost(j.dist_zero, j.bits_zero, j.lambda);
      if (zc < best_cost) {
        best_cost = zc;
        best_reco = best_resi = j.dist_zero;
        best = -1;
        best_cbf = false;
      }
    }
  }
  xvcgpu_tx_eval_result r;
  r.cost = best_cost;
  r.dist_reco = best_reco;
  r.dist_resi = best_resi;
  r.best = best;
  r.cbf = best_cbf ? 1 : 0;
  r.reserved[0] = r.reserved[1] = r.reserved[2] = 0;
  out[i] = r;
}

__global__ void __launch_bounds__(256)
root_cbf_kernel(const xvcgpu_root_cbf_job *jobs, int n, xvcgpu_root_cbf_result *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const xvcgpu_root_cbf_job j = jobs[i];
  unsigned long long resi = 0, fin = 0, zero = 0;
  bool any = false;
  for (int c = 0; c < 3; c++) {
    resi += j.dist_resi[c];
    fin += j.dist_reco[c];
    zero += j.dist_zero[c];
    any |= j.cbf[c] != 0;
  }
  bool luma_cbf = j.cbf[0] != 0;
  // :316-339 (root cbf is only a choice when some component is coded: for a CU
  // without any cbf the two sides are the same state and "zero < non-zero" is
  // decided on the numbers, as written)
  const unsigned long long cost_non_zero = rd_cost(resi, j.bits_non_zero, j.lambda);
  const unsigned long long cost_zero = rd_cost(zero, j.bits_root_zero, j.lambda);
  bool root = any;
  if (cost_zero < cost_non_zero) {
    resi = fin = zero;
    root = false;
    luma_cbf = false;
  }
  // :342-361
  bool second = false;
  if ((j.flags & XVC_CBF_FAST_SELECT) && luma_cbf) {
    const unsigned long long cost_full = rd_cost(resi, j.bits_full, j.lambda);
    // cost_full > best_cu_cost * kFastTransformSelectCostFactor, in double as written
    second = !((double)cost_full > (double)j.best_cu_cost * 1.1);
  }
  xvcgpu_root_cbf_result r;
  r.sum_dist_final = fin;
  r.sum_dist_resi = resi;
  r.root_cbf = root ? 1 : 0;
  r.second_pass = second ? 1 : 0;
  for (int k = 0; k < 6; k++) r.reserved[k] = 0;
  out[i] = r;
}

#endif  // XVCGPU_K_RD_H_
