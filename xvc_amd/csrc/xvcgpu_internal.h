// xvcgpu_internal.h -- host-side structs and device-side views shared by the
// kernels of libxvcgpu.so.  gfx950 (CDNA4, wave64) only.
#ifndef XVCGPU_INTERNAL_H_
#define XVCGPU_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/xvcgpu.h"
#include "tz_pattern.h"

// One plane of a device picture: `p` addresses sample (0,0); the replicated
// border of `border` samples lies at negative / beyond-size coordinates.
struct PlaneView {
  uint16_t *p;
  int stride;  // in samples
  int w, h;
  int border;
};

struct PicView {
  PlaneView c[3];
  int bd;
};

// The reference pictures of a batch as a table indexed by a job's slot: one launch
// covers the pictures of both lists (k_inter_pred.h, the *_refs searches).
#define XVC_MAX_REF_SLOTS 10  // 2 lists x kMaxNumRefPics (common.h:144)

struct RefTable {
  PicView pic[XVC_MAX_REF_SLOTS];
  int n;   // entries in use (the *_refs searches: a job whose slot is >= n is no job)
};

struct xvcgpu_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  hipEvent_t ev_sync;  // xvcgpu_wait_for
  // xvcgpu_set_short_kernel_priority: a second, high-priority stream for the short
  // kernels at the end of a frame pass (inverse transform, the fused tail) and the two
  // events that hand the chain over and back
  hipStream_t copy_stream;   // xvcgpu_upload_ahead: created on first use
  bool inv_one_launch;       // xvcgpu_inv_transform_batch: one workgroup-per-block launch
  hipStream_t hi_stream;
  hipEvent_t ev_hi_in, ev_hi_out;
  hipEvent_t ev_pool[64];  // xvcgpu_timer_mark slots, created on first use
  std::string err;
  // transform matrices [type 1..5][log2 size 1..6], device copy
  int16_t *d_tx_tables;
  int16_t *d_tx_tables_t;  // transposed
  // TZ candidate pattern (tz_pattern.h), device copy
  TzCand *d_tz_pattern;
  // straggler-first scheduling of the motion search (k_me2.h): three rotating
  // records of where the slow jobs sat
  struct Me2Rot *d_me_rot;
  int me_epoch;
  // per-block partial results of xvcgpu_picture_ssd
  unsigned long long *d_ssd_part;
  int ssd_part_cap;  // in blocks
  // xvcgpu_deblock_pad_ssd (k_tail.h): two words per 64x64 tile
  unsigned long long *d_tail_part;
  int tail_cap;      // in tiles
  // scratch of the whole-picture statistics passes (k_stats.h): a signed
  // histogram of 4096 buckets followed by one word per picture row (row CRCs /
  // row remainders of the dithering export)
  uint32_t *d_stats;
  int stats_rows_cap;
  struct CrcTables *d_crc_tables;  // [0]: 8-bit samples, [1]: wider; built on first use
  int *d_intra_done;               // xvcgpu_intra_recon_waves: finished jobs per dependency wave
  int intra_done_cap;
  bool rdoq_classified_proved;     // the last xvcgpu_fwd_from_me_classify_prove ran the all-zero proof itself
  int rdoq_qp_hint;                // luma QP of the last xvcgpu_fwd_from_me_classify (-1: none)
  int rdoq_prove_zero;             // quant_rdo: the all-zero proof ahead of the walk: 0 / 1 / -1 by batch size
  int intra_waves_grid;            // workgroups of its cooperative launch (0: not determined yet)
  // xvcgpu_cs_segs_launch: the launches' segment records, page-locked, two halves used in
  // turn (a half is re-used once the launches that read it have passed: seg_ring_ev)
  unsigned char *h_seg_ring;
  size_t seg_ring_pos;
  int seg_ring_half;
  hipEvent_t seg_ring_ev[2];
  bool seg_ring_used[2];
  int rdoq_four_lane_only;         // xvcgpu_quant_rdo_set_four_lane_only: the general class's launch is skipped
  int *h_rdoq_misuse;              // page-locked: set by the walk when such a batch held a general-class block
  // scratch of xvcgpu_quant_rdo_batch (k_rdoq.h): the three class lists + their
  // counters, and 26 bytes per coefficient of the batch
  int *d_rdoq_lists;
  int rdoq_lists_cap;      // in blocks
};

struct xvcgpu_picture {
  xvcgpu_ctx *ctx;
  int w, h, bd;
  void *base;
  size_t bytes;
  bool own;
  PicView v;
};

// Offsets (in int16 entries) of the transform matrices inside the packed
// table blob: index [type-1][log2 size]; filled by xvcgpu_tables.cpp.
struct TxTableLayout {
  int off[7][7];   // [type - 1][log2 size]; row 5 (XVC_TX_SKIP) unused, row 6 = XVC_TX_DCT2_LOW
  int total;
};
const TxTableLayout &xvcgpu_tx_layout();
const int16_t *xvcgpu_tx_host_tables();    // packed host copy, M[k][n]
const int16_t *xvcgpu_tx_host_tables_t();  // same layout, transposed

#endif  // XVCGPU_INTERNAL_H_
