// k_multi.h -- the kernels of a frame pass for several pictures per launch
// (xvcgpu_frame_pass_multi): grid y = picture, every kernel is the
// single-picture kernel's body run with that picture's arguments (MultiArgs,
// dev_common.h).  Same code, same results; what changes is what runs beside
// what: with k independent pictures in flight on k streams the kernels of
// different kinds meet each other (a motion search holds every wave slot while
// another picture's 20 us transform waits), here the k searches run together,
// then the k transforms, ...: kernels of one kind beside each other cost
// 59 / 45 / 18 us per picture (search / RDOQ / forward transform) against
// 77 / 122 / 22 alone (tools/throughput_cost.py).
#ifndef XVCGPU_K_MULTI_H_
#define XVCGPU_K_MULTI_H_

#include "k_me2.h"
#include "k_misc.h"
#include "k_rdoq.h"
#include "k_recon.h"
#include "k_tail.h"
#include "k_tx.h"
#include "k_tx2.h"

// recon_from_me_kernel<false, FWD>: FWD = prediction + forward transform
// (coefficients out), else the whole QuantFast reconstruction
struct ReconMultiArgs {
  PicView orig, ref, rec;
  const xvcgpu_me_block *blocks;
  const xvcgpu_me_result *results;
  int n_cus, qp_y, qp_c, ref_poc;
  int32_t *nnz_out;
  xvcgpu_cu_info *cus;
  int16_t *coeffs;
  const uint32_t *coeff_off;
};
template <bool FWD>
__global__ void __launch_bounds__(256)
recon_from_me_multi_kernel(MultiArgs<ReconMultiArgs> m, const int16_t *tx_tables,
                           const int16_t *tx_tables_t, TxTableLayout lay) {
  const ReconMultiArgs &a = m.a[blockIdx.y];
  recon_from_me_kernel_body<false, FWD>(a.orig, a.ref, a.rec, a.blocks, a.results, a.n_cus,
                                        a.qp_y, a.qp_c, 0, a.ref_poc, a.nnz_out, a.cus, tx_tables,
                                        tx_tables_t, lay, nullptr, nullptr, a.coeffs, a.coeff_off,
                                        FwdClassify());
}

// the three kernels of xvcgpu_quant_rdo_batch
struct RdoqMultiArgs {
  const xvcgpu_tx_block *blocks;
  int n;
  const int16_t *coeffs;
  const uint32_t *d_off;
  int16_t *levels;
  int32_t *nnz_out;
  RdoqLists l;
  const xvcgpu_rdoq_contexts *rq_ctx;
  const xvcgpu_rdoq_params *rq_prm;
};
__global__ void __launch_bounds__(256)
rdoq_classify_multi_kernel(MultiArgs<RdoqMultiArgs> m, int bd) {
  const RdoqMultiArgs &a = m.a[blockIdx.y];
  rdoq_classify_kernel_body(bd, a.blocks, a.n, a.coeffs, a.d_off, a.levels, a.nnz_out, a.l);
}
__global__ void __launch_bounds__(1024)
rdoq_compact_multi_kernel(MultiArgs<RdoqMultiArgs> m) {
  const RdoqMultiArgs &a = m.a[blockIdx.y];
  rdoq_compact_kernel_body(a.n, a.l);
}
__global__ void __launch_bounds__(64, RDOQ4_MIN_WAVES)
quant_rdo_packed4_multi_kernel(MultiArgs<RdoqMultiArgs> m, int bd, int g16) {
  const RdoqMultiArgs &a = m.a[blockIdx.y];
  quant_rdo_packed4_kernel_body(bd, a.blocks, a.l, g16, a.coeffs, a.d_off, a.levels, a.nnz_out,
                                a.rq_ctx, a.rq_prm, nullptr);
}
__global__ void __launch_bounds__(64, RDOQ_MIN_WAVES)
quant_rdo_packed_multi_kernel(MultiArgs<RdoqMultiArgs> m, int bd) {
  const RdoqMultiArgs &a = m.a[blockIdx.y];
  quant_rdo_packed_kernel_body(bd, a.blocks, a.l, a.coeffs, a.d_off, a.levels, a.nnz_out,
                               a.rq_ctx, a.rq_prm, nullptr);
}

// xvcgpu_inv_transform_batch (TX_MODE_INV): blocks up to 16x16 by the wave
// kernel, the rest by the scanning general-path kernel
struct InvMultiArgs {
  PicView pred, rec;
  const xvcgpu_tx_block *blocks;
  int n;
  int16_t *levels;
  const uint32_t *level_off;
  int32_t *nnz;
};
__global__ void __launch_bounds__(64 * TX2_WAVES)
inv_wave_multi_kernel(MultiArgs<InvMultiArgs> m, const int16_t *tx_tables,
                      const int16_t *tx_tables_t, TxTableLayout lay) {
  const InvMultiArgs &a = m.a[blockIdx.y];
  residual_wave_kernel_body<TX_MODE_INV, false>(a.pred, a.pred, a.rec, a.blocks, a.n, a.levels,
                                                a.level_off, a.nnz, tx_tables, tx_tables_t, lay,
                                                nullptr, nullptr, nullptr);
}
__global__ void __launch_bounds__(TX_THREADS)
inv_general_multi_kernel(MultiArgs<InvMultiArgs> m, const int16_t *tx_tables, TxTableLayout lay) {
  const InvMultiArgs &a = m.a[blockIdx.y];
  residual_kernel_body<TX_MODE_INV, false>(a.pred, a.pred, a.rec, a.blocks, a.n, a.levels,
                                           a.level_off, a.nnz, tx_tables, lay, nullptr, nullptr,
                                           nullptr);
}

struct CuInfoMultiArgs {
  const xvcgpu_me_block *blocks;
  const xvcgpu_me_result *results;
  const int32_t *nnz, *luma_tx_index;
  int n, qp_y, qp_c, ref_poc;
  xvcgpu_cu_info *cus;
};
__global__ void __launch_bounds__(256)
cu_info_multi_kernel(MultiArgs<CuInfoMultiArgs> m) {
  const CuInfoMultiArgs &a = m.a[blockIdx.y];
  cu_info_from_me_kernel_body(a.blocks, a.results, a.nnz, a.luma_tx_index, a.n, a.qp_y, a.qp_c,
                              a.ref_poc, a.cus);
}

// xvcgpu_deblock_pad_ssd with the SSD parts
struct TailMultiArgs {
  DbParams d;
  PicView src, dst;
  PlaneView orig;
  int shift, tiles;
  unsigned long long *part, *out;
};
__global__ void __launch_bounds__(256, TAIL_MIN_WAVES)
deblock_tail_multi_kernel(MultiArgs<TailMultiArgs> m) {
  const TailMultiArgs &a = m.a[blockIdx.y];   // (grid x padded to 8: the body drops the rest)
  deblock_tail_kernel_body<true>(a.d, a.src, a.dst, a.orig, a.shift, a.part);
}
__global__ void __launch_bounds__(256)
picture_ssd_sum_multi_kernel(MultiArgs<TailMultiArgs> m) {
  const TailMultiArgs &a = m.a[blockIdx.y];
  picture_ssd_sum_kernel_body(a.part, a.tiles, a.out);
}

#endif  // XVCGPU_K_MULTI_H_
