/*
 * xvc_syntax.h -- the parsed syntax of one picture: the hand-over point between
 * the (host, bit-serial, out of scope) CABAC parser and the reconstruction
 * stage that runs on the device.
 *
 * In the reference the parser (xvc_dec_lib/cu_reader.cc, syntax_reader.cc)
 * fills CodingUnit objects and CuDecoder::DecompressCu
 * (xvc_dec_lib/cu_decoder.cc:84-138) reconstructs them one by one.  Here the
 * same information is a flat array of leaf CUs in coding order: exactly the
 * fields DecompressComponent, InterPrediction::MotionCompensation,
 * IntraPrediction::Predict, Quantize::Inverse, InverseTransform::Transform and
 * DeblockingFilter read from a CodingUnit, after CalculateMV
 * (inter_prediction.cc:632-687) has resolved merge / AMVP into vectors.
 * Plain C, no HIP / torch / C++ types.
 */
#ifndef XVC_SYNTAX_H_
#define XVC_SYNTAX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XVC_CU_AFFINE 1 /* CodingUnit::GetUseAffine() */
#define XVC_CU_LIC 2    /* CodingUnit::GetUseLic()    */

/* IntraMode of xvc_cu_syntax.intra_mode[]: 0 planar, 1 DC, 2..66 angular,
 * XVC_CU_INTRA_LM = IntraMode::kLmChroma (cu_types.h:79-84) */
#define XVC_CU_INTRA_LM (-2)

typedef struct xvc_cu_syntax {
  int16_t x, y;          /* luma position                                     */
  uint8_t w, h;          /* luma size: 4..64, any power-of-two pair           */
  uint8_t tree;          /* CuTree: 0 primary (Y, or Y+U+V), 1 secondary (U+V
                          * of an intra picture, picture_data.cc:71-76)       */
  uint8_t pred_mode;     /* PredictionMode: 0 intra, 1 inter                  */
  int8_t qp[3];          /* CodingUnit::GetQp(comp): raw qp per component     */
  uint8_t inter_dir;     /* InterDir: 0 L0, 1 L1, 2 bi                        */
  uint8_t cbf[3];        /* GetCbf(comp)                                      */
  uint8_t flags;         /* XVC_CU_*                                          */
  uint8_t tx_skip[3];    /* GetTransformSkip(comp)                            */
  uint8_t reserved0;
  uint8_t tx_type[3][2]; /* GetTransformType(comp, 0 vertical / 1 horizontal):
                          * xvcgpu_tx_type values (0 = kDefault)              */
  int8_t ref_idx[2];     /* GetRefIdx(list); -1: list unused                  */
  int8_t intra_mode[3];  /* GetIntraMode(comp) (DM chroma already resolved)   */
  uint8_t reserved1;
  int32_t mv[2][3][2];   /* [list][corner][x,y] 1/16 pel: corner 0 for a
                          * translational CU, the three affine corners else   */
  uint32_t level_off[3]; /* comp's w*h levels (int16, row-major) in the
                          * picture's level array; read only when cbf[comp]   */
} xvc_cu_syntax;

/* PicturePredictionType values (picture_types.h:54-59) */
#define XVC_PIC_BI 0
#define XVC_PIC_UNI 1
#define XVC_PIC_INTRA 2

typedef struct xvc_picture_syntax {
  int32_t width, height; /* luma, internal size                               */
  int32_t bitdepth;
  int32_t poc;
  int32_t pic_type;      /* XVC_PIC_*                                          */
  int32_t deblock;       /* PictureData::GetDeblock()                          */
  int32_t beta_offset, tc_offset;
  int32_t pad_border;    /* tid == 0 || !highest layer (picture_decoder.cc:194) */
  int32_t num_ref[2];
  int32_t ref_poc[2][5]; /* ReferencePictureLists::GetRefPoc(list, idx)        */
  int32_t n_cus;
  int32_t n_levels;
} xvc_picture_syntax;

#ifdef __cplusplus
}
#endif
#endif /* XVC_SYNTAX_H_ */
