/*
 * xvcgpu_types.h -- plain-C data formats shared by the C-ABI (xvcgpu.h), the
 * HIP kernels and the CPU oracle.  No HIP / torch / C++ types here.
 *
 * Every struct restates (as a flat POD) exactly the fields that the cited
 * reference code reads on the hot path; nothing else is carried.
 */
#ifndef XVCGPU_TYPES_H_
#define XVCGPU_TYPES_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sample = uint16_t (reference: common.h:34-38, XVC_HIGH_BITDEPTH=1 build),
 * Coeff / Residual = int16_t (common.h:39-40). */
typedef uint16_t xvc_sample;
typedef int16_t xvc_coeff;

/* Distortion metric selector; numeric values follow the declaration order of
 * the reference's `enum class MetricType` (sample_metric.h:37-46). */
enum xvcgpu_metric {
  XVC_METRIC_SSD = 0,
  XVC_METRIC_SATD = 1,
  XVC_METRIC_SATD_ACONLY = 2,
  XVC_METRIC_SAD = 3,
  XVC_METRIC_SAD_FAST = 4,
  XVC_METRIC_SAD_ACONLY = 5,
  XVC_METRIC_SAD_ACONLY_FAST = 6,
  XVC_METRIC_STRUCTURAL_SSD = 7
};

/* 1-D transform type; values follow `enum class TransformType`
 * (cu_types.h): kDefault, kDct2, kDct5, kDct8, kDst1, kDst7. */
enum xvcgpu_tx_type {
  XVC_TX_DEFAULT = 0,
  XVC_TX_DCT2 = 1,
  XVC_TX_DCT5 = 2,
  XVC_TX_DCT8 = 3,
  XVC_TX_DST1 = 4,
  XVC_TX_DST7 = 5,
  /* not a TransformType: in xvcgpu_tx_block.tx_hor it selects
   * Forward/InverseTransform::TransformSkip (cu.GetTransformSkip(comp),
   * transform_encoder.cc:213-227, :270-274; blocks <= 4x4 only) */
  XVC_TX_SKIP = 6,
  /* not a TransformType either: DCT-2 of size 4 / 8 / 16 / 32 with the 6-bit
   * matrices (TransformData::kDct2Transform4..32, transform_data.cc:26-107) and
   * the stage's shift without kTransformHighPrecisionShift - what kDefault / kDct2
   * mean for those sizes under Restrictions::disable_ext2_transform_high_precision
   * (restricted mode; transform.cc:91-99, :458-605, :876-884).  The binding maps
   * the type per direction; sizes 2 and 64 and the other four types keep their
   * high-precision matrices in that mode (their shift absorbs the 2 bits), so
   * they pass unchanged. */
  XVC_TX_DCT2_LOW = 7
};

/* MvCorner order (cu_types.h / coding_unit.h:152-155 GetMvCorner). */
enum xvcgpu_mv_corner {
  XVC_CORNER_UL = 0,
  XVC_CORNER_UR = 1,
  XVC_CORNER_DL = 2,
  XVC_CORNER_DR = 3
};

/* Per-CU metadata read by the deblocking filter
 * (deblocking_filter.cc:79-241: position/size, pred mode, luma cbf, raw qp
 * per component, ref idx / ref POC per list, the four corner MVs per list).
 * The per-4x4-cell table (picture_data.h:102-107 cu_pic_table_) becomes an
 * int32 index into an array of these records (-1 = no CU, i.e. nullptr). */
typedef struct xvcgpu_cu_info {
  uint16_t x, y;       /* luma position of the CU                          */
  uint8_t w, h;        /* luma size                                        */
  uint8_t intra;       /* CodingUnit::IsIntra()                            */
  uint8_t cbf_luma;    /* CodingUnit::GetCbf(kY)                           */
  int8_t qp_y;         /* CodingUnit::GetQp(kY)  (raw qp)                  */
  int8_t qp_c;         /* CodingUnit::GetQp(kU)  (raw chroma qp)           */
  int8_t ref_idx0;     /* GetRefIdx(L0) (uni-pred pictures compare this)   */
  int8_t reserved;
  int32_t ref_poc[2];  /* GetRefPoc(list); -1 when the list is unused      */
  int32_t mv[2][4][2]; /* [list][corner][x,y] in 1/16 pel                  */
} xvcgpu_cu_info;

/* One motion-estimation job = one (CU, reference picture) pair, i.e. one call
 * of InterSearch::MotionEstNormal (inter_search.cc:606-662) with the TZ
 * search method.  Everything the reference derives from neighbouring CUs
 * (AMVP predictor, previous CU's full-pel result) is an input here. */
/* xvcgpu_me_block.fullpel_mv is a flag byte (0 / 1 as before: bit 0):
 * XVC_ME_FULLPEL_MV  cu.GetFullpelMv(): vector differences priced and the
 *                    result kept in whole samples (inter_tz_search.cc:96,
 *                    inter_search.cc:645-650);
 * XVC_ME_USE_LIC     cu.GetUseLic(): the CU tries local illumination
 *                    compensation, so the search compares with the block
 *                    means removed - kSadAcOnly[Fast] in the full-pel stage,
 *                    kSatdAcOnly in the sub-pel stage (GetFullpelMetric /
 *                    GetSubpelMetric, inter_search.cc:1059-1076). */
#define XVC_ME_FULLPEL_MV 1
#define XVC_ME_USE_LIC 2
typedef struct xvcgpu_me_block {
  int16_t x, y;        /* luma position                                    */
  uint8_t w, h;        /* luma size, each in {4,8,16,32,64}                */
  uint8_t depth_nonzero; /* cu.GetDepth() != 0 (inter_tz_search.cc:117)    */
  uint8_t fullpel_mv;  /* XVC_ME_FULLPEL_MV | XVC_ME_USE_LIC                */
  int32_t mvp_x, mvp_y;   /* AMVP predictor, 1/16 pel (unclipped)          */
  int32_t prev_x, prev_y; /* previous_fullpel_ for this (list,ref_idx)     */
  uint32_t lambda16;   /* floor(65536*sqrt(lambda)) (inter_tz_search.cc:98)*/
  int32_t search_range;/* GetSearchRangeUniPred (inter_search.cc:1050)     */
} xvcgpu_me_block;

/* Result of the full-pel + sub-pel search for one xvcgpu_me_block. */
typedef struct xvcgpu_me_result {
  int32_t fullpel_x, fullpel_y; /* TzSearch::Search return value           */
  int32_t mv_x, mv_y;           /* SubpelSearch return value, 1/16 pel     */
  uint32_t fullpel_cost;        /* state.cost_best at exit                 */
  uint32_t subpel_dist;         /* *out_dist of SubpelSearch (SATD)        */
} xvcgpu_me_result;

/* One bi-prediction refinement step for one CU (SearchBiIterative inner
 * loop, inter_search.cc:392-433): the searched list's block descriptor, the
 * MV already chosen for the OTHER list and the uni-pred MV of the searched
 * list that centres the +-4 window (mv_bootstrap, inter_search.cc:620-627). */
typedef struct xvcgpu_bi_block {
  xvcgpu_me_block blk;            /* x,y,w,h, fullpel_mv, mvp, lambda16 used   */
  int32_t other_mv_x, other_mv_y; /* MV into the other list's picture, 1/16   */
  int32_t boot_mv_x, boot_mv_y;   /* window centre, 1/16 pel                   */
} xvcgpu_bi_block;

/* One bi-pred motion-compensation job (inter_prediction.cc:710-738). */
typedef struct xvcgpu_mc_bi_block {
  int16_t x, y;        /* luma position of the CU                          */
  uint8_t w, h;        /* luma size of the CU                              */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t reserved;
  int32_t mv0_x, mv0_y; /* list-0 MV, 1/16 pel luma                        */
  int32_t mv1_x, mv1_y; /* list-1 MV                                       */
} xvcgpu_mc_bi_block;

/* One affine motion-compensation job (InterPrediction::MotionCompAffine,
 * inter_prediction.cc:1044-1136) for one component of one uni-pred CU:
 * mv[0] top-left, mv[1] top-right, mv[2] bottom-left corner, 1/16 pel. */
typedef struct xvcgpu_mc_affine_block {
  int16_t x, y;        /* luma position of the CU                          */
  uint8_t w, h;        /* luma size of the CU                              */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t reserved;
  int32_t mv[3][2];    /* [corner][x,y]                                    */
} xvcgpu_mc_affine_block;

/* One uni-pred motion compensation job with local illumination compensation
 * (InterPrediction::MotionCompensationMv with cu.GetUseLic(),
 * inter_prediction.cc:740-758 -> LocalIlluminationComp :1555-1575 ->
 * DeriveLicParams :1577-1663): the prediction is scaled / offset by a linear
 * model fitted to the row above and the column left of the block - samples of
 * the current reconstruction against the reference picture displaced by the
 * rounded full-pel MV.  The model needs the neighbouring CUs' reconstruction. */
#define XVC_LIC_HAS_ABOVE 1 /* cu.GetCodingUnitAbove() != nullptr */
#define XVC_LIC_HAS_LEFT 2  /* cu.GetCodingUnitLeft() != nullptr  */
typedef struct xvcgpu_mc_lic_block {
  int16_t x, y;        /* luma position of the CU                          */
  uint8_t w, h;        /* luma size of the CU                              */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t neighbors;   /* XVC_LIC_HAS_* bits                                */
  int32_t mv_x, mv_y;  /* 1/16 pel, before ClipMv                          */
  int16_t above_x, above_y; /* luma position of the CU above (its ClipMv)  */
  int16_t left_x, left_y;   /* luma position of the CU to the left         */
} xvcgpu_mc_lic_block;

/* One inter prediction job of the decoder = InterPrediction::MotionCompensation
 * (inter_prediction.cc:710-738) for one component of any inter CU: uni- or
 * bi-prediction (ref[list] >= 0 selects the lists: an index into the array of
 * reference pictures passed with the batch), translational (mv[list][0]) or
 * affine (XVC_INTER_AFFINE: the three corner vectors, MotionCompAffine
 * :1044-1136), with local illumination compensation (XVC_INTER_LIC: applied
 * to each list's Sample prediction, the bi-pred average then starts from the
 * compensated samples, :725-731; never together with affine, :1021-1041). */
#define XVC_INTER_AFFINE 1
#define XVC_INTER_LIC 2
typedef struct xvcgpu_inter_block {
  int16_t x, y;        /* luma position of the CU                          */
  uint8_t w, h;        /* luma size of the CU                              */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t flags;       /* XVC_INTER_*                                      */
  int8_t ref[2];       /* reference picture slot per list, -1 = unused     */
  uint8_t neighbors;   /* XVC_LIC_HAS_* (LIC only)                         */
  uint8_t reserved;
  int16_t above_x, above_y; /* LIC: luma position of the CU above / left   */
  int16_t left_x, left_y;   /*      (their ClipMv, see xvcgpu_mc_lic_block)*/
  int32_t mv[2][3][2]; /* [list][corner][x,y], 1/16 pel, before ClipMv     */
} xvcgpu_inter_block;

/* A luma position in a scratch picture (xvcgpu_inter_pred_batch_to): where the
 * job's block goes instead of the CU's own position; chroma lands at (x >> 1,
 * y >> 1), so x and y are even. */
typedef struct xvcgpu_block_pos {
  int16_t x, y;
} xvcgpu_block_pos;

/* One block copy between two pictures (xvcgpu_copy_blocks): w x h samples of
 * component `comp`, positions in samples of that component. */
typedef struct xvcgpu_copy_block {
  int16_t sx, sy;      /* in the source picture                            */
  int16_t dx, dy;      /* in the destination picture                       */
  uint8_t w, h;
  uint8_t comp;
  uint8_t reserved;
} xvcgpu_copy_block;

/* One intra prediction job = IntraPrediction::FillReferenceState + Predict
 * (intra_prediction.cc:81-147) for one component of one CU, 67-mode set.
 * Positions / sizes are in samples of `comp`.  The neighbour fields are what
 * IntraPrediction::DetermineNeighbors (:688-705) derives from the CU map:
 * which already reconstructed neighbours exist (the prediction reads the
 * reconstruction picture around the block). */
#define XVC_INTRA_HAS_ABOVE_LEFT 1
#define XVC_INTRA_HAS_ABOVE 2
#define XVC_INTRA_HAS_LEFT 4
#define XVC_INTRA_NUM_MODES 67 /* 0 planar, 1 DC, 2..66 angular (18 hor, 50 ver) */
/* chroma only: the linear model from the CU's reconstructed luma
 * (IntraMode::kLmChroma; prediction entry points, not the SATD table) */
#define XVC_INTRA_MODE_LM_CHROMA 67
typedef struct xvcgpu_intra_block {
  int16_t x, y;        /* position in the component plane                  */
  uint8_t w, h;        /* size in the component plane, {4,...,64}          */
  uint8_t comp;        /* 0 = Y (reference filtering / edge filters), 1, 2 */
  uint8_t mode;        /* IntraMode 0..66 (ignored by the all-modes search) */
  uint8_t neighbors;   /* XVC_INTRA_HAS_* bits                              */
  uint8_t above_right; /* GetCuSizeAboveRight: available samples, 0..h     */
  uint8_t below_left;  /* GetCuSizeBelowLeft: available samples, 0..w      */
  uint8_t reserved;
} xvcgpu_intra_block;

/* One residual-pipeline job = one (CU, component) pair, i.e. one call of
 * TransformEncoder::TransformAndReconstruct (transform_encoder.cc:203-285)
 * with the non-RDO quantiser. Positions/sizes are in samples of `comp`. */
typedef struct xvcgpu_tx_block {
  int16_t x, y;        /* position in the component plane                  */
  uint8_t w, h;        /* size in the component plane, {2,4,...,64}        */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t tx_hor;      /* xvcgpu_tx_type for the horizontal 1-D pass       */
  uint8_t tx_ver;      /* xvcgpu_tx_type for the vertical 1-D pass         */
  uint8_t dst4x4;      /* can_dst_4x4 (intra luma 4x4 default)             */
  int8_t qp;           /* raw qp for this component (Qp::GetQpRaw)         */
  uint8_t intra_pic;   /* XVC_TXF_* flags; 0 / 1 = inter / intra picture
                        * with the reference's defaults                    */
} xvcgpu_tx_block;

/* xvcgpu_tx_block.intra_pic bits.  The defaults (bits clear) are what the
 * reference does out of the box for an inter CU: QuantFast applies
 * sign-data hiding (CoeffSignHideFast, rdo_quant.cc:196-199, :448-573) over
 * the diagonal coefficient scan. */
#define XVC_TXF_INTRA_PIC 1       /* pic_type == kIntra: rounding offset 171/512 */
#define XVC_TXF_NO_SIGN_HIDING 2  /* Restrictions::disable_transform_sign_hiding */
#define XVC_TXF_RDOQ 16           /* quantise with RdoQuant::QuantRdo (needs the
                                   * batch's xvcgpu_rdoq_params / contexts)   */
#define XVC_TXF_SCAN_SHIFT 2      /* bits 2-3: ScanOrder of the CU, 0 diagonal,
                                   * 1 horizontal, 2 vertical
                                   * (TransformHelper::DetermineScanOrder,
                                   * transform.cc:1614-1637)                  */

/* ---- RDOQ (RdoQuant::QuantRdo, rdo_quant.cc:203-446) ---------------------- *
 * The quantiser the reference's encoder always runs (encoder_settings.h:59,
 * transform_encoder.cc:230).  It reads the entropy coder only through
 * ContextModel::GetEntropyBits of the coefficient-coding contexts of
 * writer.GetContexts() (rdo_quant.cc:254): the host snapshots those context
 * states (ContextModel::state_: (state << 1) | mps, 0..127) once per batch -
 * the extended residual context set (Contexts::coeff_ext, cabac.h:158-166;
 * the reference's default: disable_ext2_cabac_alt_residual_ctx == false). */
typedef struct xvcgpu_rdoq_contexts {
  uint8_t csbf[2][2];        /* coeff_ext.csbf_luma / csbf_chroma              */
  uint8_t sig_luma[54];      /* coeff_ext.sig_luma                             */
  uint8_t sig_chroma[12];    /* coeff_ext.sig_chroma                           */
  uint8_t greater1_luma[16]; /* coeff_ext.greater1_luma (also the greater2
                              * contexts in this set, cabac.cc:641-665)        */
  uint8_t greater1_chroma[6];
  uint8_t last_x_luma[25];   /* coeff_last_pos_x_luma                          */
  uint8_t last_y_luma[25];
  uint8_t last_x_chroma[3];
  uint8_t last_y_chroma[3];
  uint8_t cbf_luma;          /* cu_cbf_luma[0]   (intra CU, luma)              */
  uint8_t cbf_chroma;        /* cu_cbf_chroma[0] (chroma)                      */
  uint8_t root_cbf;          /* cu_root_cbf[0]   (inter CU, luma)              */
  uint8_t reserved;
} xvcgpu_rdoq_contexts;

/* Per (CU, component) inputs of QuantRdo that are double arithmetic in the
 * reference and therefore computed by the host:
 *   lambda    = (int64)(qp.GetLambdaScaled(comp) * 65536 + 0.5)   (:251-252)
 *   rd_factor = (int64)(inv_scale * inv_scale / lambda / 16 /
 *               (1 << 2 * (bitdepth - 8)) + 0.5)                  (:590-594) */
#define XVC_RDOQ_INTRA_CU 1   /* cu.IsIntra(): the luma cbf context (:756-758) */
#define XVC_RDOQ_NO_2X2 2     /* !encoder_settings.rdo_quant_2x2: 2-wide blocks
                               * take QuantFast (:208-216)                     */
typedef struct xvcgpu_rdoq_params {
  int64_t lambda;
  int64_t rd_factor;
  uint16_t ctx_index;        /* which xvcgpu_rdoq_contexts of the batch        */
  uint8_t flags;             /* XVC_RDOQ_*                                     */
  uint8_t reserved[5];
} xvcgpu_rdoq_params;

/* ---- the bits of a CU's inter prediction syntax ------------------------------ *
 * InterSearch::GetInterPredBits as the encoder runs it by default
 * (inter_search.cc:1131-1135, fast_inter_pred_bits == 0): the candidate's
 * CuWriter::WriteInterPrediction (cu_writer.cc:122-172) through a throw-away
 * RdoSyntaxWriter - every context-coded bin costs
 * ContextModel::kEntropyBits_[state ^ bin] and moves its context on
 * (entropy_encoder.cc:44-51), bypass bins cost 1 << 15, the result is the sum
 * >> 15.  The entropy coder enters only through the states of the eleven contexts
 * the syntax of ONE CU reads: the host snapshots them (ContextModel::state_) when
 * it hands the CU over; the neighbour / size dependent selections
 * (cabac.cc:361-371, :464-489) are made by the host. */
#define XVC_ICTX_PIC_BI 1      /* bi-predictive picture: the inter direction is coded */
#define XVC_ICTX_CAN_AFFINE 2  /* cu.CanUseAffine(): the affine flag is coded          */
#define XVC_ICTX_PIC_LIC 4     /* the picture uses local illumination compensation    */
typedef struct xvcgpu_inter_contexts {
  uint8_t merge_flag;    /* inter_merge_flag[0]                                       */
  uint8_t inter_dir_bi;  /* GetInterDirBiCtx(cu)                                      */
  uint8_t inter_dir_l;   /* inter_dir[4]                                              */
  uint8_t affine_flag;   /* GetAffineCtx(cu)                                          */
  uint8_t ref_idx[2];    /* inter_ref_idx[0..1]                                       */
  uint8_t mvd[2];        /* inter_mvd[0..1]                                           */
  uint8_t mvp_idx;       /* inter_mvp_idx[0]                                          */
  uint8_t fullpel_mv;    /* GetInterFullpelMvCtx(cu)                                  */
  uint8_t lic_flag;      /* lic_flag[0]                                               */
  uint8_t flags;         /* XVC_ICTX_*                                                */
  uint8_t num_refs[2];   /* pictures per list                                         */
  uint16_t frac_bits;    /* bitstream_writer.GetFractionalBits(): the throw-away writer
                          * starts from the live coder's fraction of a bit
                          * (syntax_writer.cc:861-865, entropy_encoder.cc:33-38)      */
} xvcgpu_inter_contexts;

/* ---- one SearchMotion of a CU, chained on the device --------------------------- *
 * InterSearch::SearchMotion (inter_search.cc:199-259) is a chain: per list and
 * picture EvalStartMvp -> search -> EvalFinalMvpIdx -> GetInterPredBits -> cost fold
 * (SearchRefIdx :456-578), then SearchBiIterative (:392-433) from the two winners,
 * then the three-way choice (:247-257); CompressInter (:74-98) runs it a second time
 * with the affine model and keeps the cheaper result.  The searches are the batched
 * entry points; what sits BETWEEN them - a few dozen integer operations on their
 * results - are the three folds below, so that the chain needs no host round trip:
 * each fold reads the previous step's results and writes the next step's jobs.
 * A pass describes what the encoder's control code knows before the search. */
#define XVC_CS_MAX_REFS 3      /* pictures per list (default_num_ref_pics 2, placebo 3) */
#define XVC_CS_FULLPEL 1       /* cu.GetFullpelMv()                                   */
#define XVC_CS_FORCE_L1_MVD_ZERO 2 /* pic_data.GetForceBipredL1MvdZero() (only-back-reference
                                * pictures, inter_search.cc:410-413, :496-518): the folds
                                * do NOT run this variant - such a pass is answered
                                * XVC_CS_WHICH_UNSUPPORTED (InterSearch::SearchMotionMultiBatch
                                * of the C++ host layer handles it)                      */
#define XVC_CS_LIC 4           /* cu.GetUseLic() (never with XVC_CS_AFFINE): the flag in the
                                * candidates' syntax, the refinement jobs the uni fold writes
                                * are to be searched with xvcgpu_bipred_search_lic (the CU's
                                * xvcgpu_mc_lic_block per slot), XVC_INTER_LIC in the evaluation's
                                * prediction jobs (their neighbour fields are the caller's).
                                * EvalStartMvp's distortions of such a pass are those of
                                * COMPENSATED predictions (inter_search.cc:980: XVC_INTER_LIC
                                * jobs of xvcgpu_inter_pred_batch_to + xvcgpu_eval_dist_batch) */
#define XVC_CS_AFFINE 8        /* the affine pass (MotionVector3, MotionEstAffine)      */
/* xvcgpu_cs_result::which of a pass the folds do not run: XVC_CS_FORCE_L1_MVD_ZERO, or
 * bi_iterations > 1 (the folds run ONE SearchBiIterative iteration on the list that lost,
 * encoder_settings.cc:58-96 default / faster / slow; placebo's 4, :34, is the host
 * layer's loop).  No refinement or evaluation job of such a pass is filled in. */
#define XVC_CS_WHICH_UNSUPPORTED 255
typedef struct xvcgpu_cs_pass {
  int16_t x, y;                /* luma position of the CU                              */
  uint8_t w, h;
  uint8_t flags;               /* XVC_CS_*                                             */
  uint8_t num_refs[2];         /* pictures per list                                    */
  int8_t same_poc_in_l0[XVC_CS_MAX_REFS]; /* list-1 picture -> the list-0 ref_idx of the
                                * same picture (its result is re-used, :536-542), -1   */
  uint32_t lambda16;           /* floor(65536 * sqrt(lambda))                          */
  xvcgpu_inter_contexts ictx;  /* what GetInterPredBits reads of the entropy coder      */
  int32_t mvp[2][XVC_CS_MAX_REFS][2][3][2]; /* GetMvpList[Affine]: [list][ref_idx][predictor]
                                * [corner][x, y]; a plain vector uses corner 0         */
  int32_t uni_job[2][XVC_CS_MAX_REFS];   /* index of the search's job / result in the
                                * xvcgpu_me_block (affine pass: xvcgpu_affine_me_block)
                                * array; -1: no search (re-used picture)               */
  int32_t start_dist[2][XVC_CS_MAX_REFS]; /* index of predictor 0's EvalStartMvp
                                * distortion (SAD of its prediction); predictor 1's
                                * follows                                             */
  int32_t prev_job[2][XVC_CS_MAX_REFS];  /* plain pass: the earlier search job of the same
                                * chain whose full-pel result is this search's
                                * previous_fullpel_ (:640-641), -1: the job holds it   */
  int32_t bi_job;              /* first of the pass's 2 * MAX_REFS * MAX_REFS refinement
                                * job slots [searched list][ref_idx][other ref_idx]    */
  int32_t plain_pass;          /* affine pass: the CU's plain pass (bootstrap vectors,
                                * the result it has to beat), else -1                  */
  int32_t eval;                /* the evaluation (3 xvcgpu_inter_block) that takes the
                                * chosen motion, -1: none                              */
  int8_t slot[2][XVC_CS_MAX_REFS]; /* reference picture slot per (list, ref_idx)       */
  uint8_t bi_iterations;       /* encoder_settings.bipred_refinement_iterations; 0 or 1:
                                * one iteration, more: XVC_CS_WHICH_UNSUPPORTED          */
  uint8_t reserved;
} xvcgpu_cs_pass;

/* What a pass computed: every intermediate the reference's loop holds (readable by
 * the host for checking) and the motion state SearchMotion ends with. */
typedef struct xvcgpu_cs_result {
  uint8_t start_idx[2][XVC_CS_MAX_REFS];  /* EvalStartMvp                              */
  uint8_t mvp_idx[2][XVC_CS_MAX_REFS];    /* EvalFinalMvpIdx                           */
  int32_t mv[2][XVC_CS_MAX_REFS][3][2];   /* uni-directional result per picture        */
  uint32_t dist[2][XVC_CS_MAX_REFS], bits[2][XVC_CS_MAX_REFS], cost[2][XVC_CS_MAX_REFS];
  uint32_t cost_list[2], cost_l1_unique;
  int8_t best_ref[2], best_ref_l1_unique; /* SearchRefIdx's winners (-1: none)         */
  uint8_t search_list;                    /* the list SearchBiIterative searched        */
  uint8_t bi_mvp_idx[XVC_CS_MAX_REFS];
  uint8_t bi_valid;                       /* a refinement was run                       */
  int32_t bi_mv[XVC_CS_MAX_REFS][3][2];
  uint32_t bi_dist[XVC_CS_MAX_REFS], bi_bits[XVC_CS_MAX_REFS], bi_cost[XVC_CS_MAX_REFS];
  /* SearchMotion's result (inter_search.cc:247-257) */
  uint8_t which;               /* 0 bi, 1 list 0, 2 list 1 (unique picture)            */
  uint8_t inter_dir;
  int8_t ref_idx[2];
  uint8_t out_mvp_idx[2];
  uint8_t zero_mvd;            /* CodingUnit::HasZeroMvd()                              */
  uint8_t chosen;              /* affine pass: 1 = this pass beat the plain one (:85-93);
                                * plain pass: 1                                        */
  uint32_t best_cost;
  int32_t out_mv[2][3][2];
  int32_t out_mvd[2][2][2];
} xvcgpu_cs_result;

/* ---- the merge ranking of a CU, folded on the device ---------------------------- *
 * InterSearch::SearchMergeCandidates (inter_search.cc:165-197): the SATD of the five
 * merge candidates' luma predictions (xvcgpu_inter_pred_batch_to + xvcgpu_metric_batch)
 * -> cost = dist + bits * lambda_sqrt in double (bits = idx + 1, the last index one
 * less), a stable sort, and the count of candidates within kFastMergeCostFactor = 1.25
 * of the cheapest (at most kFastMergeNumCand = 4).  xvcgpu_cs_merge_fold does that
 * between the launches and writes the motion of ranked candidate i < num into the
 * position's i-th EVALUATION SLOT (3 xvcgpu_inter_block: Y, U, V - geometry and
 * component filled in by the caller; the fold sets flags, ref and mv), ref = -1 / -1
 * in the slots from num on: CuEncoder::CompressMerge's loop (cu_encoder.cc:598-628)
 * can then be enqueued - every ranked candidate's CompressAndEvalCbf - without the
 * host having seen the ranking. */
#define XVC_CS_MERGE_CANDS 5   /* constants::kNumInterMergeCandidates */
#define XVC_CS_MERGE_SLOTS 4   /* InterSearch::kFastMergeNumCand      */
typedef struct xvcgpu_cs_merge {
  double lambda_sqrt;          /* qp.GetLambdaSqrt()                                     */
  int32_t dist;                /* index of candidate 0's distortion (uint64; 1..4 follow) */
  int32_t cand;                /* index of candidate 0's xvcgpu_inter_block (the luma job
                                * the ranking predicted with; 1..4 follow)              */
  int32_t slot;                /* the position's first evaluation slot (slot s = blocks
                                * 3 s .. 3 s + 2 of the evaluation array), -1: none      */
  int32_t reserved;
} xvcgpu_cs_merge;
typedef struct xvcgpu_cs_merge_result {
  double cost[XVC_CS_MERGE_CANDS];   /* sorted                                           */
  int32_t order[XVC_CS_MERGE_CANDS]; /* out_cand_list: merge index per rank               */
  int32_t num;                       /* candidates to evaluate                           */
  int32_t reserved[2];
} xvcgpu_cs_merge_result;

/* ---- the steps of many pictures' CU-state chains in one launch -------------------- *
 * xvcgpu_cs_segs_launch (xvcgpu.h): a SEGMENT is the argument list of one of the batched
 * entry points a chain issues - p[] in that entry point's own order, the pictures from
 * the chain's environment (xvcgpu_cs_env_create) - and a launch runs up to hundreds of
 * segments of one kind, grid y = segment. */
#define XVC_CS_SEG_MC_METRIC_REFS 0  /* p: cands, out, slots                  n jobs          */
#define XVC_CS_SEG_START_FOLD 1      /* p: passes, start_dist, me_jobs, me_res, aff_jobs; i0 = first pass, n passes */
#define XVC_CS_SEG_UNI_FOLD 2        /* p: passes, me_res, aff_res, bi_jobs, aff_jobs; i0, n     */
#define XVC_CS_SEG_BI_FOLD 3         /* p: passes, bi_res, aff_res, ev_inter; i0, n              */
#define XVC_CS_SEG_MERGE_FOLD 4      /* p: merges, dist, cands, results, ev_inter; i0, n         */
#define XVC_CS_SEG_ME_REFS 5         /* p: blocks, results, slots; i0 = block class (16/32/64)  */
#define XVC_CS_SEG_BI_REFS 6         /* p: jobs, results, slots; i0 = block class               */
#define XVC_CS_SEG_AFFINE_REFS 7     /* p: blocks, results, slots; i0 = CU height (16/32/64)    */
#define XVC_CS_SEG_INTER_PRED 8      /* p: blocks, dst positions (into the env's s_pred)        */
#define XVC_CS_SEG_RESIDUAL_AT 9     /* p: tx blocks, level offsets, nnz, contexts, params, source
                                      * positions, evaluation candidates (or 0), their output;
                                      * r0 = candidates in front (xvcgpu_residual_rdoq_batch_at) */
#define XVC_CS_SEG_EVAL_DIST 10      /* p: candidates (orig_at), out                            */
#define XVC_CS_SEG_FETCH 11          /* p: source (device), destination (page-locked host memory
                                      * the device can write: xvcgpu_host_alloc); n bytes, a
                                      * multiple of 4, both 4-byte aligned                      */
#define XVC_CS_SEG_KINDS 12
typedef struct xvcgpu_cs_env xvcgpu_cs_env;
typedef struct xvcgpu_cs_seg {
  int32_t n, i0, r0, r1;
  uint64_t p[8];               /* device addresses                                        */
  const xvcgpu_cs_env *env;
} xvcgpu_cs_seg;

/* One distortion of an evaluation (xvcgpu_eval_dist_batch): what CompressAndEvalCbf /
 * CompressAndEvalTransform compare per component and alternative - the prediction
 * against the original (the cbf-zero distortion, transform_encoder.cc:116-117) or a
 * reconstruction against it (:284) - with the component's distortion weight.  One
 * launch prices all of them, whatever their component. */
typedef struct xvcgpu_eval_cand {
  int16_t x, y;        /* block position in the component plane (all three pictures) */
  uint8_t w, h;
  uint8_t metric;      /* xvcgpu_metric                                               */
  int8_t qp;           /* raw luma qp (structural SSD)                                */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                                         */
  uint8_t versus;      /* 0: the prediction picture, 1: the reconstruction picture    */
  int16_t ox, oy;      /* orig_at != 0: the block's position in `orig` (component
                        * plane) - the original picture itself instead of a copy of
                        * the block beside the candidate's slot                       */
  uint8_t orig_at;
  uint8_t reserved;
  double weight;       /* Qp::GetDistortionWeight(comp)                               */
} xvcgpu_eval_cand;

/* One motion-compensation job (InterPrediction::MotionCompensationMv,
 * inter_prediction.cc:740-758) for one component of one uni-pred CU. */
typedef struct xvcgpu_mc_block {
  int16_t x, y;        /* luma position of the CU                          */
  uint8_t w, h;        /* luma size of the CU                              */
  uint8_t comp;        /* 0 = Y, 1 = U, 2 = V                              */
  uint8_t reserved;
  int32_t mv_x, mv_y;  /* 1/16 pel luma MV (clipped inside, ClipMv)        */
} xvcgpu_mc_block;

#ifdef __cplusplus
}
#endif
/* One picture's worth of the hot path in a single call (xvcgpu_frame_pass):
 * the launches a host issues per picture - search, CompressAndEvalCbf,
 * deblocking, PadBorder, PSNR parts - selected by `phases`, in this order.
 * All pointers are device memory laid out as for the individual entry points. */
#define XVC_FP_ENCODE 1     /* xvcgpu_me_search_sized + xvcgpu_recon_from_me    */
#define XVC_FP_DEBLOCK_V 2  /* xvcgpu_deblock_rows pass 0 on [db_y_begin, db_y_end) */
#define XVC_FP_DEBLOCK_H 4  /* xvcgpu_deblock_rows pass 1 on [db_y_begin, dbh_y_end) */
#define XVC_FP_PAD 8        /* xvcgpu_pad_border(rec)                           */
#define XVC_FP_SSD 16       /* xvcgpu_picture_ssd_rows(orig, rec, luma)         */
typedef struct xvcgpu_frame_pass_args {
  const struct xvcgpu_picture *orig, *ref;
  struct xvcgpu_picture *rec;
  const xvcgpu_me_block *d_me;   /* the own CUs' search jobs (CUs <= 16x16)     */
  xvcgpu_me_result *d_results;
  int32_t n_cus;                 /* own CUs                                     */
  int32_t max_block_size;
  int32_t qp_y, qp_c, ref_poc;
  int32_t *d_nnz;                /* 3 per own CU                                */
  xvcgpu_cu_info *d_cus_own;     /* metadata of the own CUs (inside d_cus)      */
  const xvcgpu_cu_info *d_cus;   /* whole picture's CU array                    */
  int32_t n_cus_total;
  const int32_t *d_cu_map;
  int32_t map_stride;
  int32_t db_y_begin, db_y_end;  /* luma rows of the vertical-edge pass         */
  int32_t dbh_y_end;             /* end row of the horizontal-edge pass         */
  int32_t ssd_y_begin, ssd_y_end;
  int32_t shift_bitdepth;        /* xvcgpu_picture_ssd_rows                     */
  uint64_t *d_ssd;
  /* quantiser of XVC_FP_ENCODE: NULL = QuantFast; else RdoQuant::QuantRdo with
   * d_rdoq_params[3 * cu + comp] and the context snapshots they index */
  const struct xvcgpu_rdoq_contexts *d_rdoq_contexts;
  const struct xvcgpu_rdoq_params *d_rdoq_params;
  /* RDOQ, throughput form (pred != NULL): the residual pipeline split around
   * the packed quantiser - xvcgpu_mc_from_me -> xvcgpu_fwd_transform_batch ->
   * xvcgpu_quant_rdo_batch -> xvcgpu_inv_transform_batch ->
   * xvcgpu_cu_info_from_me - with these work buffers (3 transform blocks per
   * own CU: Y, U, V; d_rdoq_params indexed like d_tx) */
  struct xvcgpu_picture *pred;
  const xvcgpu_tx_block *d_tx;
  const uint32_t *d_level_off;
  const int32_t *d_luma_tx_index;
  int16_t *d_coeffs, *d_levels;
  int32_t n_tx;
  uint32_t n_coeffs;
  /* optional, only for pictures whose CUs are all at least 8x8: a picture of
   * rec's size.  A call that runs ENCODE, DEBLOCK_V, DEBLOCK_H, PAD and SSD
   * over all rows then writes the unfiltered reconstruction here and ends with
   * ONE launch, xvcgpu_deblock_pad_ssd(scratch_rec -> rec), instead of five */
  struct xvcgpu_picture *scratch_rec;
  /* RDOQ, throughput form: the caller's word that d_tx holds no block of the quantiser's
   * general class (xvcgpu_quant_rdo_set_four_lane_only, for this call's batch): pictures
   * of CUs 8x8 .. 16x16 coded with the diagonal scan, say */
  int32_t tx_four_lane_only;
  /* the caller's word that every job of d_me is a 16x16 or a 16x8 CU (a picture whose width
   * is a multiple of 16 on the 16-sample CU grid): XVCGPU_ME_ONLY_SQ16 for this call's search */
  int32_t me_only_sq16;
} xvcgpu_frame_pass_args;

/* One job of xvcgpu_affine_me_batch: InterSearch::MotionEstAffine for one
 * (list, ref_idx) of a CU (inter_search.cc:664-749).  Vectors are {x, y} in
 * 1/16 pel; mvp / bootstrap / other_mv are MotionVector3 (top-left, top-right,
 * bottom-left corner).  XVC_AFFINE_ME_BIPRED: the search of SearchBiIterative
 * (:394-435) - the target is 2 * orig - the affine prediction of the other
 * list (other_mv on the `ref_other` picture; SubtractWeighted), distortions
 * are halved, 5 iterations, the bootstrap vector always wins. */
#define XVC_AFFINE_ME_HAS_BOOTSTRAP 1
#define XVC_AFFINE_ME_BIPRED 2
typedef struct xvcgpu_affine_me_block {
  int16_t x, y;        /* luma position of the CU */
  uint8_t w, h;        /* 16, 32 or 64 each (CodingUnit::CanUseAffine: > 8) */
  uint8_t flags;       /* XVC_AFFINE_ME_HAS_BOOTSTRAP | XVC_AFFINE_ME_BIPRED */
  uint8_t reserved;
  uint32_t lambda16;   /* floor(65536 * sqrt(lambda)) */
  int32_t mvp[3][2];
  int32_t bootstrap[3][2];
  int32_t other_mv[3][2]; /* XVC_AFFINE_ME_BIPRED: the other list's vectors */
} xvcgpu_affine_me_block;

typedef struct xvcgpu_affine_me_result {
  int32_t mv[3][2];    /* best_mv */
  uint32_t dist;       /* *out_dist: SATD of the best prediction (>> 1 with BIPRED) */
  uint32_t iterations; /* gradient iterations that produced a non-zero update */
} xvcgpu_affine_me_result;

/* ---- C1, decision half: the folds of CompressAndEvalTransform and of
 * CompressAndEvalCbf (transform_encoder.cc:53-201, inter_search.cc:261-365).
 * Distortions come from the device's own kernels; the bits an alternative costs
 * come from the caller's entropy coder (a per-alternative input).
 *
 * One xvcgpu_tx_eval_job = one CompressAndEvalTransform call for a (CU,
 * component): its alternatives, in the reference's evaluation order, occupy
 * [alt_first, alt_first + n_alt) of the batch's alternative arrays.  Every cost
 * is dist_resi + (Cost)(bits * lambda + 0.5) in double as the reference forms it;
 * an alternative replaces the best so far only when strictly cheaper. */
#define XVC_TXE_KIND_NORMAL 0   /* default transform (:97-110)                     */
#define XVC_TXE_KIND_TSKIP 1    /* transform skip (:146-162)                       */
#define XVC_TXE_KIND_SELECT 2   /* a transform-select index (:164-194)             */
#define XVC_TXE_CBF_ZERO 1      /* TxSearchFlags::kCbfZero: after the NORMAL
                                 * alternative, if it kept a coefficient, the
                                 * all-zero block competes (:112-144)              */
#define XVC_TXE_FAST_SELECT 2   /* fast_transform_select_eval with kCbfZero: the
                                 * SELECT alternatives are skipped when the best so
                                 * far has no coefficient (:176-180)               */
#define XVC_TXE_PREV_CBF 4      /* second pass: the state prev_cost belongs to has
                                 * a coefficient                                   */
#define XVC_TXE_DIST_INVALID 0xffffffffffffffffull /* TransformAndReconstruct returned
                                 * max(): the alternative breaks a signalling
                                 * invariant (:239-252)                            */
typedef struct xvcgpu_tx_eval_alt {
  uint64_t dist_reco;      /* TransformAndReconstruct's return value               */
  uint64_t dist_resi;      /* the distortion the cost is formed with (= dist_reco
                            * unless fast_inter_transform_dist applies, :71-83)     */
  uint32_t bits;           /* rdo_writer.GetNumWrittenBits() for the alternative    */
  uint8_t kind;            /* XVC_TXE_KIND_*                                        */
  uint8_t cbf;             /* it kept a coefficient                                 */
  uint8_t reserved[2];
} xvcgpu_tx_eval_alt;
typedef struct xvcgpu_tx_eval_job {
  double lambda;           /* qp.GetLambda()                                        */
  uint64_t prev_cost;      /* *prev_cost of the second pass; ~0: first pass         */
  uint64_t dist_zero;      /* prediction against the original (XVC_TXE_CBF_ZERO)    */
  uint32_t bits_zero;      /* WriteCbf(cu, comp, false)                             */
  uint32_t alt_first;
  uint8_t n_alt;
  uint8_t flags;           /* XVC_TXE_*                                             */
  uint8_t reserved[6];
} xvcgpu_tx_eval_job;
/* best: index of the winning alternative relative to alt_first; -1: the all-zero
 * block; -2: nothing beat prev_cost (the previous state stays) */
typedef struct xvcgpu_tx_eval_result {
  uint64_t cost, dist_reco, dist_resi;
  int32_t best;
  uint8_t cbf;             /* the winning state has a coefficient                   */
  uint8_t reserved[3];
} xvcgpu_tx_eval_result;

/* One CompressAndEvalCbf tail per CU (inter_search.cc:316-361): the root-cbf-zero
 * test on the components' winning states and the gate of the transform-select
 * second pass. */
#define XVC_CBF_FAST_SELECT 1   /* encoder_settings.fast_transform_select_eval      */
typedef struct xvcgpu_root_cbf_job {
  double lambda;
  uint64_t dist_resi[3], dist_reco[3], dist_zero[3];  /* per component              */
  uint64_t best_cu_cost;   /* the RD search's best cost for this CU so far          */
  uint32_t bits_non_zero;  /* GetCuBitsResidual of the winning states               */
  uint32_t bits_root_zero; /* WriteRootCbf(false)                                   */
  uint32_t bits_full;      /* GetCuBitsFull of the winning states                   */
  uint8_t cbf[3];          /* the winning states' cbf                               */
  uint8_t flags;           /* XVC_CBF_*                                             */
} xvcgpu_root_cbf_job;
typedef struct xvcgpu_root_cbf_result {
  uint64_t sum_dist_final, sum_dist_resi;
  uint8_t root_cbf;        /* 0: the zero-residual CU won (every cbf cleared)       */
  uint8_t second_pass;     /* the transform-select pass is to be evaluated          */
  uint8_t reserved[6];
} xvcgpu_root_cbf_result;

/* One device-to-device copy of xvcgpu_copy_segments (row slabs of a picture
 * to / from the staging buffer of a multi-GPU exchange). */
typedef struct xvcgpu_copy_segment {
  const void *src;
  void *dst;
  uint64_t bytes;
} xvcgpu_copy_segment;

#endif /* XVCGPU_TYPES_H_ */
