/*
 * xvc_oracle.h -- CPU restatement ("oracle") of the xvc per-CU hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the CPU baseline.  The product
 * path (xvc_amd/, libxvcgpu.so) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_
 * vs_ref.py, this container only) against the reference's own compiled code
 * (oracle/_ref/libxvcref.so, built by oracle/Makefile from /root/reference
 * sources where they lie) on seeded random inputs, and against the committed
 * golden vectors in tests/golden/ (generated from that same reference build
 * by tools/gen_golden.py).  The reference has no known-answer vectors of its
 * own for this path (its tests are relational).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/src).
 */
#ifndef XVC_ORACLE_H_
#define XVC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/xvcgpu_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- M1..M7 distortion metrics (xvc_enc_lib/sample_metric.cc:171-223) ---- *
 * Returns static_cast<Distortion>(dist * weight) exactly as Compare() does.
 * qp_raw_y / structural_strength only matter for XVC_METRIC_STRUCTURAL_SSD. */
uint64_t xo_metric_ss(int metric, int bitdepth, int qp_raw_y,
                      int structural_strength, double weight, int w, int h,
                      const uint16_t *s1, ptrdiff_t st1,
                      const uint16_t *s2, ptrdiff_t st2);
/* Residual (int16) x Sample overload (sample_metric.cc:226-277). */
uint64_t xo_metric_rs(int metric, int bitdepth, int qp_raw_y,
                      int structural_strength, double weight, int w, int h,
                      const int16_t *s1, ptrdiff_t st1,
                      const uint16_t *s2, ptrdiff_t st2);
/* Residual x Residual SSD (sample_metric.cc:279-298). */
uint64_t xo_ssd_rr(int bitdepth, double weight, int w, int h,
                   const int16_t *s1, ptrdiff_t st1,
                   const int16_t *s2, ptrdiff_t st2);
/* SampleMetric::ComparePicture with kSsd over one plane, and the sample
 * count used by ComputePsnr (sample_metric.cc:37-155): per-64x64-block SSD
 * each right-shifted by 2*(bd-8). Returns the summed distortion. */
uint64_t xo_picture_ssd(int bitdepth, int w, int h,
                        const uint16_t *p1, ptrdiff_t st1,
                        const uint16_t *p2, ptrdiff_t st2,
                        uint64_t *psnr_dist, uint64_t *psnr_samples);
/* ... restricted to the blocks whose first row is in [y_begin, y_end). */
uint64_t xo_picture_ssd_rows(int bitdepth, int w, int h, int y_begin, int y_end,
                             const uint16_t *p1, ptrdiff_t st1, const uint16_t *p2,
                             ptrdiff_t st2, uint64_t *psnr_dist,
                             uint64_t *psnr_samples);

/* ---- I1/I2 interpolation (xvc_common_lib/inter_prediction.cc) ---- */
/* MotionCompUniPred -> Sample (inter_prediction.cc:1138-1154). `ref` points
 * at the full-pel position; frac in 1/16 (luma) or 1/32 (chroma) units. */
void xo_mc_uni(int bitdepth, int is_chroma, int w, int h, int frac_x,
               int frac_y, const uint16_t *ref, ptrdiff_t ref_stride,
               uint16_t *pred, ptrdiff_t pred_stride);
/* MotionCompUniPred -> int16 14-bit intermediate (:1156-1172). */
void xo_mc_uni_bipred(int bitdepth, int is_chroma, int w, int h, int frac_x,
                      int frac_y, const uint16_t *ref, ptrdiff_t ref_stride,
                      int16_t *pred, ptrdiff_t pred_stride);
/* AddAvgBi (inter_prediction.cc:1540-1553, sample_buffer.h:89-106). */
void xo_add_avg(int bitdepth, int w, int h, const int16_t *s1, ptrdiff_t st1,
                const int16_t *s2, ptrdiff_t st2, uint16_t *dst,
                ptrdiff_t dst_stride);
/* ClipMv (inter_prediction.cc:769-782). mv in/out 1/16 pel. */
void xo_clip_mv(int pos_x, int pos_y, int pic_w, int pic_h, int *mv_x,
                int *mv_y);
/* MotionCompensationMv for one component of a uni-pred CU
 * (inter_prediction.cc:740-758 + GetFullpelRef :1174-1205). Planes point at
 * sample (0,0) of a padded picture. (x,y,w,h) are LUMA position/size. */
void xo_mc_block(int bitdepth, int comp, int x, int y, int w, int h, int mv_x,
                 int mv_y, int pic_w, int pic_h, const uint16_t *ref_plane,
                 ptrdiff_t ref_stride, uint16_t *pred, ptrdiff_t pred_stride);

/* ---- X1..X3, Q1 transforms and (de)quantisation ---- */
/* ForwardTransform::Transform (transform.cc:869-961), high-precision mode. */
void xo_fwd_transform(int bitdepth, int w, int h, int tx_hor, int tx_ver,
                      int dst4x4, const int16_t *resi, ptrdiff_t resi_stride,
                      int16_t *coeff, ptrdiff_t coeff_stride);
/* InverseTransform::Transform (transform.cc:83-182). */
void xo_inv_transform(int bitdepth, int w, int h, int tx_hor, int tx_ver,
                      int dst4x4, int dc_only, const int16_t *coeff,
                      ptrdiff_t coeff_stride, int16_t *resi,
                      ptrdiff_t resi_stride);
/* TransformSkip forward / inverse (transform.cc:963-995, :184-215). */
void xo_fwd_transform_skip(int bitdepth, int w, int h, const int16_t *resi,
                           ptrdiff_t resi_stride, int16_t *coeff,
                           ptrdiff_t coeff_stride);
void xo_inv_transform_skip(int bitdepth, int w, int h, const int16_t *coeff,
                           ptrdiff_t coeff_stride, int16_t *resi,
                           ptrdiff_t resi_stride);
/* Quantize::Inverse (quantize.cc:94-125). qp_raw = Qp::GetQpRaw(comp). */
void xo_dequant(int bitdepth, int qp_raw, int w, int h, const int16_t *in,
                ptrdiff_t in_stride, int16_t *out, ptrdiff_t out_stride);
/* RdoQuant::QuantFast (rdo_quant.cc:156-201); returns the number of non-zero
 * levels.  xo_quant_fast = the disable_transform_sign_hiding path;
 * xo_quant_fast2 = the reference's default QuantFast: with sign-data hiding
 * (CoeffSignHideFast, rdo_quant.cc:448-573) when sign_hide != 0; scan_order
 * 0 diagonal (every inter CU), 1 horizontal, 2 vertical (transform.cc:1614-1637) */
int xo_quant_fast2(int bitdepth, int qp_raw, int intra_pic, int sign_hide,
                   int scan_order, int w, int h, const int16_t *in, ptrdiff_t in_stride,
                   int16_t *out, ptrdiff_t out_stride);
int xo_quant_fast(int bitdepth, int qp_raw, int intra_pic, int w, int h,
                  const int16_t *in, ptrdiff_t in_stride, int16_t *out,
                  ptrdiff_t out_stride);
/* TransformAndReconstruct with QuantFast (transform_encoder.cc:203-285):
 * resi = orig - pred; fwd; quant; [dequant; inv; rec = clip(pred + resi')]
 * or rec = pred when cbf == 0.  Writes levels to `coeff_out` (stride w).
 * Returns num_non_zero. */
int xo_residual_pipeline(int bitdepth, const xvcgpu_tx_block *blk,
                         const uint16_t *orig, ptrdiff_t orig_stride,
                         const uint16_t *pred, ptrdiff_t pred_stride,
                         uint16_t *rec, ptrdiff_t rec_stride,
                         int16_t *coeff_out);

/* ---- D1..D4 deblocking (xvc_common_lib/deblocking_filter.cc:56-450) ---- */
/* planes[c] point at sample (0,0). cu_map: one int32 per 4x4 luma cell,
 * row-major with `map_stride` entries per row, -1 = no CU. */
void xo_deblock_picture(int bitdepth, int pic_w, int pic_h, int pic_is_bipred,
                        int beta_offset, int tc_offset, int subblock_size,
                        const xvcgpu_cu_info *cus, const int32_t *cu_map,
                        int map_stride, uint16_t *const planes[3],
                        const ptrdiff_t strides[3]);

void xo_deblock_picture_planes(int bitdepth, int pic_w, int pic_h, int pic_is_bipred,
                               int beta_offset, int tc_offset, int subblock_size,
                               const xvcgpu_cu_info *cus, const int32_t *cu_map,
                               int map_stride, uint16_t *const planes[3],
                               const ptrdiff_t strides[3], int comp_mask);

void xo_deblock_rows(int bitdepth, int pic_w, int pic_h, int pic_is_bipred,
                     int beta_offset, int tc_offset, int subblock_size,
                     const xvcgpu_cu_info *cus, const int32_t *cu_map,
                     int map_stride, uint16_t *const planes[3],
                     const ptrdiff_t strides[3], int pass, int y_begin,
                     int y_end);

/* ---- P1 border extension (xvc_common_lib/yuv_pic.cc:118-150) ---- */
void xo_pad_border(int w, int h, int border_x, int border_y, uint16_t *plane,
                   ptrdiff_t stride);

/* ---- T1..T3 motion search ---- */
/* DetermineMinMaxMv (inter_prediction.cc:801-817) in full-pel units. */
void xo_min_max_mv(int pos_x, int pos_y, int pic_w, int pic_h, int center_x,
                   int center_y, int search_range, int mv_min[2],
                   int mv_max[2]);
/* TzSearch::Search (inter_tz_search.cc:84-171) with kSad / kSadFast chosen by
 * GetFullpelMetric (inter_search.cc:1059-1069), eval_prev_mv_search_result
 * on. Planes point at (0,0) of padded pictures. */
void xo_tz_search(int bitdepth, const xvcgpu_me_block *blk, int pic_w,
                  int pic_h, const uint16_t *orig, ptrdiff_t orig_stride,
                  const uint16_t *ref, ptrdiff_t ref_stride, int out_mv[2],
                  uint32_t *out_cost);
/* InterSearch::FullSearch (inter_search.cc:853-891) on an int16 target
 * (bi-pred "2*orig - pred_other", 64-stride buffer). */
void xo_full_search(int bitdepth, int x, int y, int w, int h, int fullpel_mv,
                    int mvp_x, int mvp_y, uint32_t lambda16, const int mv_min[2],
                    const int mv_max[2], const int16_t *target,
                    ptrdiff_t target_stride, const uint16_t *ref,
                    ptrdiff_t ref_stride, int out_mv[2]);
/* InterSearch::SubpelSearch (inter_search.cc:893-949) with kSatd. */
void xo_subpel_search(int bitdepth, const xvcgpu_me_block *blk, int pic_w,
                      int pic_h, const uint16_t *orig, ptrdiff_t orig_stride,
                      const uint16_t *ref, ptrdiff_t ref_stride,
                      const int fullpel[2], int out_mv[2], uint32_t *out_dist);
/* One bi-prediction refinement step: target = 2*orig - MC(other list)
 * (SearchBiIterative, inter_search.cc:392-433), +-4 FullSearch on the int16
 * target (:853-891), SubpelSearch with TOrig = Residual, distortion halved
 * (MotionEstNormal, :606-662). */
void xo_bipred_search(int bitdepth, const xvcgpu_bi_block *job, int pic_w,
                      int pic_h, const uint16_t *orig, ptrdiff_t orig_stride,
                      const uint16_t *ref_other, ptrdiff_t other_stride,
                      const uint16_t *ref_search, ptrdiff_t search_stride,
                      xvcgpu_me_result *out);
/* I3 (affine half): InterPrediction::MotionCompAffine -> Sample
 * (inter_prediction.cc:1044-1136): per-sub-block MVs from the three corner
 * MVs mv[0..2] = {x,y} in 1/16 pel; x,y,w,h = luma geometry of the CU. */
void xo_mc_affine_block(int bitdepth, int comp, int x, int y, int w, int h,
                        const int mv[3][2], int pic_w, int pic_h,
                        const uint16_t *ref_plane, ptrdiff_t ref_stride,
                        uint16_t *pred, ptrdiff_t pred_stride);
/* InterSearch::GetSubpelDist (inter_search.cc:951-964): MotionCompensationMv
 * of a luma block, then CompareSample(orig, prediction). */
uint64_t xo_mc_metric(int bitdepth, int metric, int qp_raw_y, int strength, int x,
                      int y, int w, int h, int mv_x, int mv_y, int pic_w, int pic_h,
                      const uint16_t *orig, ptrdiff_t orig_stride,
                      const uint16_t *ref, ptrdiff_t ref_stride);
/* MotionCompensation for a bi-pred CU, one component
 * (inter_prediction.cc:710-738: two 14-bit predictions + AddAvgBi). */
void xo_mc_bipred_block(int bitdepth, int comp, int x, int y, int w, int h,
                        int mv0_x, int mv0_y, int mv1_x, int mv1_y, int pic_w,
                        int pic_h, const uint16_t *ref0, ptrdiff_t rs0,
                        const uint16_t *ref1, ptrdiff_t rs1, uint16_t *pred,
                        ptrdiff_t pred_stride);
/* MotionCompensationMv of a CU with local illumination compensation
 * (inter_prediction.cc:740-758, :1555-1663). */
void xo_mc_lic_block(int bitdepth, const xvcgpu_mc_lic_block *b, int pic_w, int pic_h,
                     const uint16_t *ref, ptrdiff_t rs, const uint16_t *rec, ptrdiff_t cs,
                     uint16_t *pred, ptrdiff_t ps);
/* T5: InterSearch::AffineGradientSearch (inter_search.cc:751-851) and
 * MotionEstAffine (:664-749; uni-pred and the bi-pred refinement search);
 * xvc_oracle_affine_me.c. */
void xo_affine_gradient_search(int width, int height, const uint16_t *pred, ptrdiff_t ps,
                               const int16_t *err, ptrdiff_t es, int mvd[4]);
void xo_affine_me(int bd, const xvcgpu_affine_me_block *b, int pic_w, int pic_h,
                  const uint16_t *orig, ptrdiff_t os, const uint16_t *ref, ptrdiff_t rs,
                  const uint16_t *ref_other, ptrdiff_t ros, xvcgpu_affine_me_result *out);
/* GetMvdBitsFullpel / GetMvdBits / GetNumExpGolombBits
 * (inter_search.cc:1150-1188). */
uint32_t xo_mvd_bits_fullpel(int mvp_x, int mvp_y, int fx, int fy,
                             int mvd_down_shift);
uint32_t xo_mvd_bits(int mvp_x, int mvp_y, int mv_x, int mv_y,
                     int mvd_down_shift);

/* ---- the hot-path frame pass (xvc_oracle_frame.c; DESIGN.md "Frame pass").
 * The same argument block drives xo_frame_pass (this restatement) and
 * xr_frame_pass (oracle/ref_harness.cc: the same composition executed by the
 * reference's own classes). ---- */
typedef struct xo_frame_args {
  int bd, pic_w, pic_h;
  int n_cus;
  const xvcgpu_me_block *me_blocks; /* one per CU */
  int n_tx;
  const xvcgpu_tx_block *tx_blocks; /* luma + chroma blocks of all CUs */
  const int32_t *luma_tx_index;     /* per CU: index of its luma tx block */
  const int32_t *cu_map;
  int map_stride;
  int qp_y, qp_c, ref_poc;
  int beta_offset, tc_offset, subblock;
  int border[3]; /* border available around every plane (>= 80/40) */
  const uint16_t *orig[3];
  ptrdiff_t orig_stride[3];
  const uint16_t *ref[3]; /* padded reference picture */
  ptrdiff_t ref_stride[3];
  uint16_t *pred[3]; /* work picture */
  ptrdiff_t pred_stride[3];
  uint16_t *rec[3]; /* output: reconstructed, deblocked, padded */
  ptrdiff_t rec_stride[3];
  xvcgpu_me_result *me_results; /* out, n_cus */
  int32_t *nnz;                 /* out, n_tx */
  xvcgpu_cu_info *cus;          /* in/out: whole picture's CU array; the own
                                   CUs [cu_base, cu_base + n_cus) are written */
  int cu_base;
  int encode_only;              /* stop before deblock / pad / SSD */
  uint64_t ssd[2];              /* out: luma SSD as ComputePsnr sums it, samples */
  int threads;                  /* <= 1: serial; else OpenMP threads for the
                                   per-CU loops (results do not depend on it) */
  /* quantiser: NULL = QuantFast for every block; else the blocks flagged
   * XVC_TXF_RDOQ take RdoQuant::QuantRdo with rdoq_params[tx index] */
  const xvcgpu_rdoq_contexts *rdoq_contexts;
  const xvcgpu_rdoq_params *rdoq_params;
  double rdoq_lambda;           /* the luma lambda the host derived the params
                                   from (xr_frame_pass builds its Qp with it and
                                   checks the fixed-point params against its own) */
} xo_frame_args;
void xo_frame_pass(xo_frame_args *a);

/* ---- whole-picture passes around the hot path (xvc_oracle_stats.c; SURVEY
 * 8f N4): sample conversion in / out (resample.cc:152-262, :304-338,
 * :475-551), CRC (checksum.cc:46-92), AQP variance (cu_encoder.cc:308-363),
 * LIC histogram distance (picture_encoder.cc:230-281). ---- */
void xo_import_plane(int in_bitdepth, int out_bitdepth, int in_w, int in_h,
                     int out_w, int out_h, const uint8_t *src,
                     ptrdiff_t src_stride_bytes, uint16_t *dst,
                     ptrdiff_t dst_stride);
void xo_export_plane(int src_bitdepth, int out_bitdepth, int dither, int w, int h,
                     const uint16_t *src, ptrdiff_t src_stride, uint8_t *out);
int xo_picture_crc(int bitdepth, int mode, int w, int h,
                   const uint16_t *const planes[3], const ptrdiff_t strides[3],
                   uint8_t *hash);
void xo_variance_map(int w, int h, const uint16_t *luma, ptrdiff_t stride,
                     uint64_t *out);
uint64_t xo_ctu_variance(int w, int h, int x, int y, int ctu_size,
                         const uint64_t *var_map);
int xo_aqp_delta_qp(uint64_t ctu_variance, int bitdepth, int aqp_strength);
int64_t xo_histogram_distance(int bitdepth, int w, int h, const uint16_t *a,
                              ptrdiff_t sa, const uint16_t *b, ptrdiff_t sb);
int xo_allow_lic(int64_t histogram_distance, int w, int h);

/* ---- intra prediction + SATD mode pre-selection (xvc_oracle_intra.c; SURVEY
 * 8f N3): IntraPrediction (xvc_common_lib/intra_prediction.cc:81-147,
 * :342-558, :688-871) and the distortions of
 * IntraSearch::DetermineSlowIntraModes (xvc_enc_lib/intra_search.cc:189-305),
 * 67-mode set. ---- */
#define XO_INTRA_REF_STRIDE 129
void xo_intra_ref_samples(int bitdepth, int w, int h, int neighbors, int above_right,
                          int below_left, const uint16_t *src, ptrdiff_t stride,
                          uint16_t *ref, uint16_t *ref_filtered);
int xo_intra_use_filtered(int w, int h, int mode);
void xo_intra_predict(int bitdepth, int is_luma, int mode, int w, int h,
                      const uint16_t *ref, const uint16_t *ref_filtered, uint16_t *out,
                      ptrdiff_t os);
void xo_intra_lm_chroma(int bitdepth, int x, int y, int w, int h, const uint16_t *luma,
                        ptrdiff_t ls, const uint16_t *chroma, ptrdiff_t cs, uint16_t *out,
                        ptrdiff_t os);
void xo_intra_pred_block(int bitdepth, const xvcgpu_intra_block *b, const uint16_t *rec,
                         ptrdiff_t rs, uint16_t *pred, ptrdiff_t ps);
void xo_intra_satd_modes(int bitdepth, const xvcgpu_intra_block *b, const uint16_t *orig,
                         ptrdiff_t os, const uint16_t *rec, ptrdiff_t rs, uint32_t *dist);

/* ---- Q2: RdoQuant::QuantRdo + CoeffSignHideRdo (xvc_oracle_rdoq.c;
 * rdo_quant.cc:203-446, :575-687) with the extended residual context set. ---- */
int xo_quant_rdo(int bd, int qp_raw, int comp, int scan_order, int sign_hide, int w, int h,
                 const xvcgpu_rdoq_contexts *ctx, const xvcgpu_rdoq_params *prm,
                 const int16_t *src, ptrdiff_t is, int16_t *out, ptrdiff_t os);
const uint32_t *xo_entropy_bits_table(void);
int xo_residual_pipeline_rdoq(int bd, const xvcgpu_tx_block *b, const xvcgpu_rdoq_contexts *ctx,
                              const xvcgpu_rdoq_params *prm, const uint16_t *orig, ptrdiff_t os,
                              const uint16_t *pred, ptrdiff_t ps, uint16_t *rec, ptrdiff_t rs,
                              int16_t *coeff_out);

/* ---- decoder reconstruction (xvc_oracle_dec.c) ---- */
#include "../include/xvc_syntax.h"
void xo_inter_pred_block(int bd, const xvcgpu_inter_block *b, int pic_w, int pic_h,
                         const uint16_t *const *ref_planes, const uint16_t *const *rec_planes,
                         uint16_t *const *pred_planes, const ptrdiff_t *strides);
void xo_intra_neighbors(const int32_t *cells, int stride, int rows, const xvc_cu_syntax *cu,
                        int comp, uint8_t *flags, uint8_t *above_right, uint8_t *below_left);
void xo_decode_picture(const xvc_picture_syntax *ps, const xvc_cu_syntax *cus,
                       const int16_t *levels, const uint16_t *const *ref_planes,
                       const int32_t ref_slot[2][5], uint16_t *const planes[3],
                       const ptrdiff_t strides[3], int border, uint8_t *nb_out,
                       uint16_t *const pre_planes[3]);

/* Transform matrix access (transform_data.cc, high-precision tables) for
 * table-equality tests: returns pointer to N*N int16 row-major, or NULL. */
const int16_t *xo_transform_matrix(int tx_type, int size);

#ifdef __cplusplus
}
#endif
#endif /* XVC_ORACLE_H_ */
