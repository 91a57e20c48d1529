/* xvc_inter_bits.h -- the bits of a CU's inter-prediction syntax, as the reference
 * encoder prices a motion candidate by default.
 *
 * InterSearch::GetInterPredBits (xvc_enc_lib/inter_search.cc:1082-1137) with
 * fast_inter_pred_bits == 0 (encoder_settings.h:85; only restricted mode sets 1):
 *     RdoSyntaxWriter rdo_writer(bitstream_writer, 0);
 *     cu_writer_.WriteInterPrediction(cu, kY, &rdo_writer);
 *     return rdo_writer.GetNumWrittenBits();
 * The RDO writer has no bit sink: EntropyEncoder::EncodeBin adds
 * ContextModel::kEntropyBits_[state ^ bin] to a 15-bit fixed-point sum and moves
 * the context to its next state, EncodeBypass adds 1 << 15
 * (entropy_encoder.cc:40-51, :78-98); GetNumWrittenBits is the sum >> 15
 * (entropy_encoder.h:43-45).  What is written: CuWriter::WriteInterPrediction
 * (cu_writer.cc:122-172) -> SyntaxWriter::WriteMergeFlag, WriteInterDir,
 * WriteAffineFlag, WriteInterRefIdx, WriteInterMvd, WriteInterMvpIdx,
 * WriteInterFullpelMvFlag, WriteLicFlag (syntax_writer.cc:52-61, :359-441, :549-563).
 *
 * The entropy coder's state enters through eleven context bytes and the live
 * coder's current fraction of a bit, which the throw-away writer inherits
 * (xvcgpu_inter_contexts, include/xvcgpu_types.h), snapshotted by the host when a
 * CU state is handed to the device.  One definition serves the C++ host layer
 * (xvc_amd/host) and the HIP kernels (csrc/k_cu_state.h): the qualifier macro
 * XVC_BITS_FN is the only thing that differs; the two tables are parameters
 * (the device keeps them in constant memory).
 *
 * The context state machine (context_model.cc:39-73): state byte = (p << 1) | mps;
 * on the MPS p moves up to 62 (bytes 124 / 125 stay, 126 / 127 are the terminate
 * state and stay); on the LPS p = trans_idx_lps[p], p == 0 flips the MPS. */
#ifndef XVC_INTER_BITS_H_
#define XVC_INTER_BITS_H_

#include <stdint.h>

#include "xvcgpu_types.h"

#ifndef XVC_BITS_FN
#define XVC_BITS_FN static inline
#endif

/* The motion candidate whose syntax is priced: the CU's inter state as
 * SearchRefIdx has set it when it calls GetInterPredBits (inter_search.cc:556-563). */
typedef struct xvc_inter_syntax {
  uint8_t inter_dir;         /* InterDir: 0 = L0, 1 = L1, 2 = bi                  */
  uint8_t use_affine;
  uint8_t fullpel_mv;
  uint8_t use_lic;
  int8_t ref_idx[2];
  uint8_t mvp_idx[2];
  uint8_t force_mvd_zero[2]; /* CodingUnit::GetForceMvdZero(list)                  */
  uint8_t reserved[2];
  int32_t mvd[2][2][2];      /* [list][corner 0 / 1 (affine only)][x, y]           */
} xvc_inter_syntax;

/* H.265 table 9-41 transIdxLps, the LPS transition of the 64-state machine the
 * reference's 128-entry kNextStateLps_ spells out (checked entry for entry against
 * the reference's table by tests/test_inter_bits.py). */
#define XVC_TRANS_IDX_LPS_LIST                                                          \
  0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, \
      21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, \
      33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63

typedef struct xvc_bits_tables {
  const uint32_t *entropy_bits;   /* ContextModel::kEntropyBits_[128]              */
  const uint8_t *trans_idx_lps;   /* [64]                                          */
} xvc_bits_tables;

XVC_BITS_FN uint8_t xvc_ctx_next(uint8_t s, int bin, const uint8_t *trans_idx_lps) {
  const int p = s >> 1, mps = s & 1;
  if (bin == mps) return (uint8_t)(s < 124 ? s + 2 : s);
  if (p == 0) return (uint8_t)(s ^ 1);
  if (p == 63) return s;
  return (uint8_t)((trans_idx_lps[p] << 1) | mps);
}

/* EncodeBin on the RDO writer */
XVC_BITS_FN void xvc_bits_bin(uint64_t *frac, uint8_t *ctx, int bin, const xvc_bits_tables *t) {
  *frac += t->entropy_bits[*ctx ^ bin];
  *ctx = xvc_ctx_next(*ctx, bin, t->trans_idx_lps);
}

/* SyntaxWriter::WriteExpGolomb (syntax_writer.cc:818-832): the number of bypass bins */
XVC_BITS_FN int xvc_exp_golomb_bins(uint32_t abs_level, uint32_t k) {
  int num_bins = 0;
  while (abs_level >= (1u << k)) {
    num_bins++;
    abs_level -= 1u << k;
    k++;
  }
  return num_bins + 1 + (int)k;
}

/* SyntaxWriter::WriteInterMvd (syntax_writer.cc:379-413) */
XVC_BITS_FN void xvc_bits_mvd(uint64_t *frac, uint8_t *ctx_mvd, int mx, int my,
                              const xvc_bits_tables *t) {
  const uint32_t ax = (uint32_t)(mx < 0 ? -mx : mx), ay = (uint32_t)(my < 0 ? -my : my);
  xvc_bits_bin(frac, &ctx_mvd[0], ax != 0, t);
  xvc_bits_bin(frac, &ctx_mvd[0], ay != 0, t);
  if (ax) xvc_bits_bin(frac, &ctx_mvd[1], ax > 1, t);
  if (ay) xvc_bits_bin(frac, &ctx_mvd[1], ay > 1, t);
  int bypass = 0;
  if (ax) bypass += (ax > 1 ? xvc_exp_golomb_bins(ax - 2, 1) : 0) + 1;
  if (ay) bypass += (ay > 1 ? xvc_exp_golomb_bins(ay - 2, 1) : 0) + 1;
  *frac += (uint64_t)bypass << 15;
}

/* GetInterPredBits of a non-merge candidate (the RD search's motion candidates;
 * a merge candidate's syntax is the merge flag + index, priced by its caller). */
XVC_BITS_FN uint32_t xvc_inter_pred_bits(const xvcgpu_inter_contexts *snapshot,
                                         const xvc_inter_syntax *s, const xvc_bits_tables *t) {
  xvcgpu_inter_contexts c = *snapshot;   /* the throw-away writer's own context copy */
  uint64_t frac = c.frac_bits & 32767u;
  xvc_bits_bin(&frac, &c.merge_flag, 0, t);                       /* WriteMergeFlag(false) */
  if (c.flags & XVC_ICTX_PIC_BI) {                                /* WriteInterDir */
    xvc_bits_bin(&frac, &c.inter_dir_bi, s->inter_dir == 2, t);
    if (s->inter_dir != 2) xvc_bits_bin(&frac, &c.inter_dir_l, s->inter_dir == 1, t);
  }
  if (c.flags & XVC_ICTX_CAN_AFFINE) xvc_bits_bin(&frac, &c.affine_flag, s->use_affine != 0, t);
  for (int l = 0; l < 2; l++) {
    if (s->inter_dir != 2 && s->inter_dir != l) continue;
    const int num_refs = c.num_refs[l];
    int ref_idx = s->ref_idx[l];
    if (num_refs > 1) {                                           /* WriteInterRefIdx */
      xvc_bits_bin(&frac, &c.ref_idx[0], ref_idx != 0, t);
      if (ref_idx && num_refs > 2) {
        ref_idx--;
        xvc_bits_bin(&frac, &c.ref_idx[1], ref_idx != 0, t);
        if (ref_idx)
          for (int i = 1; i < num_refs - 2; i++) {
            frac += 1u << 15;
            if (i == ref_idx) break;
          }
      }
    }
    if (!s->force_mvd_zero[l]) {
      xvc_bits_mvd(&frac, c.mvd, s->mvd[l][0][0], s->mvd[l][0][1], t);
      if (s->use_affine) xvc_bits_mvd(&frac, c.mvd, s->mvd[l][1][0], s->mvd[l][1][1], t);
    }
    /* WriteInterMvpIdx: WriteUnaryMaxSymbol(idx, max 1) = one bin */
    xvc_bits_bin(&frac, &c.mvp_idx, s->mvp_idx[l] != 0, t);
  }
  /* CodingUnit::HasZeroMvd (coding_unit.cc:445-453): the first mvd of the used lists */
  int zero;
  if (s->inter_dir == 2)
    zero = !(s->mvd[0][0][0] | s->mvd[0][0][1] | s->mvd[1][0][0] | s->mvd[1][0][1]);
  else
    zero = !(s->mvd[s->inter_dir][0][0] | s->mvd[s->inter_dir][0][1]);
  if (!zero && !s->use_affine) xvc_bits_bin(&frac, &c.fullpel_mv, s->fullpel_mv != 0, t);
  if ((c.flags & XVC_ICTX_PIC_LIC) && !s->use_affine)
    xvc_bits_bin(&frac, &c.lic_flag, s->use_lic != 0, t);
  return (uint32_t)(frac >> 15);
}

#endif /* XVC_INTER_BITS_H_ */
