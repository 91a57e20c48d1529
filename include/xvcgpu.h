/*
 * xvcgpu.h -- C-ABI of libxvcgpu.so: the MI355X (gfx950) implementation of the
 * data-parallel part of xvc's per-CU encoder inner loop.
 *
 * This is the drop-in boundary.  The reference
 * has no device boundary: its "operator plug-in point" is three structs of
 * C function pointers called once per block (sample_metric.h:166-191,
 * inter_prediction.h:176-216) plus direct member calls for transforms,
 * dequant, deblocking and padding.  One block per call is tens of ns - far
 * below a kernel launch - so every entry point here is the BATCHED form of
 * one of those calls: same arithmetic, same argument meaning, N independent
 * blocks per launch.  The reference-side binding a maintainer would add is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; no HIP/torch types.  Device memory is passed as
 *    `void *` device pointers obtained from xvcgpu_malloc() (or any HIP
 *    allocation of the same device, e.g. a torch tensor's data_ptr()).
 *  - every function returns an xvcgpu_status; xvcgpu_last_error() gives text.
 *    No exceptions cross the ABI (mirrors xvc_enc_return_code, xvcenc.h:34-45).
 *  - all launches go to the context's stream (xvcgpu_set_stream) and are
 *    asynchronous; xvcgpu_sync() waits.  Calls on one context must come from
 *    one thread at a time (the reference API is single-caller too,
 *    xvcenc.h:153-190); use one context per worker for concurrency, exactly
 *    as the reference uses one PictureEncoder per worker thread
 *    (thread_encoder.cc:99-159).
 *  - there is NO CPU fallback: without a gfx950 device xvcgpu_create() fails.
 */
#ifndef XVCGPU_H_
#define XVCGPU_H_

#include "xvcgpu_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum xvcgpu_status {
  XVCGPU_OK = 0,
  XVCGPU_INVALID_ARGUMENT = 10, /* cf. XVC_ENC_INVALID_ARGUMENT, xvcenc.h:37 */
  XVCGPU_NO_DEVICE = 20,
  XVCGPU_OUT_OF_MEMORY = 30,
  XVCGPU_DEVICE_ERROR = 40,
  XVCGPU_UNSUPPORTED = 50
} xvcgpu_status;

typedef struct xvcgpu_ctx xvcgpu_ctx;         /* one per device/worker      */
typedef struct xvcgpu_picture xvcgpu_picture; /* padded 4:2:0 planes in HBM */

/* ---- context ------------------------------------------------------------ */
xvcgpu_status xvcgpu_create(int device, xvcgpu_ctx **out);
void xvcgpu_destroy(xvcgpu_ctx *ctx);
const char *xvcgpu_last_error(const xvcgpu_ctx *ctx);
/* Version string "xvcgpu <major>.<minor> gfx950". Callable without a GPU. */
const char *xvcgpu_version(void);
/* The hipStream_t the context launches on (its own stream unless
 * xvcgpu_set_stream replaced it) - for callers that must order other work
 * after it, e.g. wrap it as torch.cuda.ExternalStream so that RCCL operations
 * are issued on the same stream as the kernels. */
void *xvcgpu_get_stream(const xvcgpu_ctx *ctx);
/* Launch on an external hipStream_t (e.g. torch's current stream, so kernels
 * order with torch copies and RCCL collectives); NULL = the device's default
 * stream.  xvcgpu_use_own_stream() returns to a private non-blocking stream
 * (the state after xvcgpu_create). */
xvcgpu_status xvcgpu_set_stream(xvcgpu_ctx *ctx, void *hip_stream);
xvcgpu_status xvcgpu_use_own_stream(xvcgpu_ctx *ctx);
/* Several contexts on one device = several queues the GPU drains concurrently
 * (the mirror of the reference's picture worker threads, thread_encoder.cc:
 * 99-159).  xvcgpu_use_priority_stream gives the context a private stream of
 * high (non-zero) or low priority: the low-priority queue fills the slots the
 * high-priority one leaves idle while its kernels ramp up and drain.
 * xvcgpu_wait_for(ctx, other): work queued on `ctx` after this call starts
 * only when everything queued on `other` before it has finished (an event). */
xvcgpu_status xvcgpu_use_priority_stream(xvcgpu_ctx *ctx, int high);
xvcgpu_status xvcgpu_wait_for(xvcgpu_ctx *ctx, xvcgpu_ctx *other);
xvcgpu_status xvcgpu_sync(xvcgpu_ctx *ctx);
/* on != 0: the short kernels that end a whole-picture xvcgpu_frame_pass (inverse
 * transform, the fused deblock / pad / PSNR tail) run on a second, high-priority
 * stream of the context, ordered with the rest of the pass by events - for several
 * picture chains in flight on one device, whose searches otherwise keep every CU
 * busy while another chain's 100-us tail waits.  Results do not depend on it. */
xvcgpu_status xvcgpu_set_short_kernel_priority(xvcgpu_ctx *ctx, int on);
/* HIP-event stopwatch on the context's stream (bench.py's timed region). */
xvcgpu_status xvcgpu_timer_begin(xvcgpu_ctx *ctx);
xvcgpu_status xvcgpu_timer_end(xvcgpu_ctx *ctx, float *elapsed_ms);
/* Several intervals in flight: record event `slot` (0..63) on the context's
 * stream now; later ask for the time between two recorded slots (waits for
 * slot_b only).  For timing one kernel while other streams keep the device
 * busy (bench.py's in-flight roofline figures). */
xvcgpu_status xvcgpu_timer_mark(xvcgpu_ctx *ctx, int slot);
xvcgpu_status xvcgpu_timer_between(xvcgpu_ctx *ctx, int slot_a, int slot_b,
                                   float *elapsed_ms);

/* ---- events and the RCCL exchange (xvc_amd/csrc/xvcgpu_comm.hip) ---------- *
 * Multi-GPU: one process and one communicator per GPU.  The ranks either code
 * independent pictures of a sub-GOP and ship each finished, padded reference
 * picture to the ranks that list it (the reference's only parallelism,
 * thread_encoder.cc:99-159; schedule: xvc_amd/host/xvc_picture_schedule.h), or
 * own CTU rows of one picture and trade the rows around a shard boundary for
 * the in-loop filter (deblocking_filter.cc:59-62).  Transfers are ncclSend /
 * ncclRecv on the communicator's own stream; events order them with the
 * kernels of the contexts:
 *   xvcgpu_event_record(ctx, ev)        ev = everything queued on ctx so far
 *   xvcgpu_comm_wait_event(comm, ev)    later transfers start after ev
 *   xvcgpu_comm_record_event(comm, ev)  ev = every transfer queued so far
 *   xvcgpu_event_wait(ctx, ev)          later kernels of ctx start after ev
 * Sends and receives that are enqueued in one agreed order on all ranks pair
 * up without deadlock; put transfers that may cross between
 * xvcgpu_comm_group_begin / _end (ncclGroupStart / ncclGroupEnd).
 * xvcgpu_comm_unique_id is called on one rank, its 128 bytes are handed to
 * the others by whatever started the processes (bench.py: torch.distributed's
 * store), then every rank calls xvcgpu_comm_create (collective). */
typedef struct xvcgpu_event xvcgpu_event;
typedef struct xvcgpu_comm xvcgpu_comm;
#define XVCGPU_COMM_ID_BYTES 128
xvcgpu_status xvcgpu_event_create(xvcgpu_ctx *ctx, xvcgpu_event **out);
void xvcgpu_event_destroy(xvcgpu_event *ev);
xvcgpu_status xvcgpu_event_record(xvcgpu_ctx *ctx, xvcgpu_event *ev);
xvcgpu_status xvcgpu_event_wait(xvcgpu_ctx *ctx, xvcgpu_event *ev);
xvcgpu_status xvcgpu_event_synchronize(xvcgpu_event *ev);
/* An upload that does not queue behind the context's kernels: on the context's COPY
 * stream (created on first use), after `after` (may be NULL: at once) - `done` is
 * recorded behind it; xvcgpu_event_wait(ctx, done) makes later kernels see the bytes.
 * The next picture's job lists go up while this picture's kernels run
 * (xvc_gpu::PictureDecoder, two device staging buffers taking turns). */
xvcgpu_status xvcgpu_upload_ahead(xvcgpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes,
                                  xvcgpu_event *after, xvcgpu_event *done);
xvcgpu_status xvcgpu_comm_unique_id(uint8_t id[XVCGPU_COMM_ID_BYTES]);
xvcgpu_status xvcgpu_comm_create(xvcgpu_ctx *ctx, const uint8_t id[XVCGPU_COMM_ID_BYTES],
                                 int world, int rank, xvcgpu_comm **out);
void xvcgpu_comm_destroy(xvcgpu_comm *comm);
int xvcgpu_comm_world(const xvcgpu_comm *comm);
int xvcgpu_comm_rank(const xvcgpu_comm *comm);
xvcgpu_status xvcgpu_comm_wait_event(xvcgpu_comm *comm, xvcgpu_event *ev);
xvcgpu_status xvcgpu_comm_record_event(xvcgpu_comm *comm, xvcgpu_event *ev);
xvcgpu_status xvcgpu_comm_sync(xvcgpu_comm *comm);
xvcgpu_status xvcgpu_comm_group_begin(xvcgpu_comm *comm);
xvcgpu_status xvcgpu_comm_group_end(xvcgpu_comm *comm);
/* The whole padded picture (all three planes with their borders: 7.8 MB at
 * 1080p, one message). */
xvcgpu_status xvcgpu_comm_send_picture(xvcgpu_comm *comm, const xvcgpu_picture *pic, int dst);
xvcgpu_status xvcgpu_comm_recv_picture(xvcgpu_comm *comm, xvcgpu_picture *pic, int src);
/* Rows [y0, y1) in luma rows (even; chroma planes move rows y0/2 .. y1/2) of
 * the components in comp_mask (bit c), with their horizontal borders: one
 * message per plane. */
xvcgpu_status xvcgpu_comm_send_rows(xvcgpu_comm *comm, const xvcgpu_picture *pic, int comp_mask,
                                    int y0, int y1, int dst);
xvcgpu_status xvcgpu_comm_recv_rows(xvcgpu_comm *comm, xvcgpu_picture *pic, int comp_mask, int y0,
                                    int y1, int src);
/* Any contiguous device range (CU metadata rows of a shard boundary). */
xvcgpu_status xvcgpu_comm_send_bytes(xvcgpu_comm *comm, const void *d_src, size_t bytes, int dst);
xvcgpu_status xvcgpu_comm_recv_bytes(xvcgpu_comm *comm, void *d_dst, size_t bytes, int src);
/* Sum over the ranks of n 64-bit counters in device memory (PSNR parts of row
 * shards), in place. */
xvcgpu_status xvcgpu_comm_all_reduce_sum_u64(xvcgpu_comm *comm, uint64_t *d_values, int n);

/* ---- recorded call sequences (HIP graphs) --------------------------------- *
 * The per-picture sequence of launches is short kernels (8-110 us each), so
 * the launch path matters.  xvcgpu_record_begin() puts the context's private
 * stream into capture: the xvcgpu_* calls that follow are recorded instead of
 * executed (no call that synchronises - sync, timer_end, free, upload,
 * download - may be made while recording); xvcgpu_record_end() returns the
 * recording as a replayable handle.  xvcgpu_replay() enqueues the whole
 * sequence with one submission; the recorded device pointers and pictures
 * must stay alive and are used as they were passed.  This has no counterpart
 * in the reference (it has no launch cost); it is the device-side analogue of
 * one PictureEncoder::Encode call (picture_encoder.cc:75-160). */
typedef struct xvcgpu_recording xvcgpu_recording;
xvcgpu_status xvcgpu_record_begin(xvcgpu_ctx *ctx);
xvcgpu_status xvcgpu_record_end(xvcgpu_ctx *ctx, xvcgpu_recording **out);
xvcgpu_status xvcgpu_replay(xvcgpu_ctx *ctx, xvcgpu_recording *rec);
void xvcgpu_recording_destroy(xvcgpu_recording *rec);

/* ---- raw device memory -------------------------------------------------- */
xvcgpu_status xvcgpu_malloc(xvcgpu_ctx *ctx, size_t bytes, void **dev_ptr);
xvcgpu_status xvcgpu_free(xvcgpu_ctx *ctx, void *dev_ptr);
xvcgpu_status xvcgpu_memcpy_h2d(xvcgpu_ctx *ctx, void *dst, const void *src,
                                size_t bytes);
xvcgpu_status xvcgpu_memcpy_d2h(xvcgpu_ctx *ctx, void *dst, const void *src,
                                size_t bytes);
xvcgpu_status xvcgpu_memset(xvcgpu_ctx *ctx, void *dst, int value, size_t bytes);
/* Page-locked host memory and the copy that does not wait: the upload is queued
 * on the context's stream (ordered with its kernels) and the call returns at
 * once - `src` must stay untouched until an event recorded after the call has
 * completed (xvcgpu_event_record / _synchronize).  With pageable `src` the
 * runtime stages the bytes itself and the call may block. */
/* (hipHostMalloc: the memory is also addressable by the device under the same pointer -
 * a kernel's small results can be written straight into it, visible to the host once
 * the stream has been synchronised: no copy kernel, no separate read-back.) */
xvcgpu_status xvcgpu_host_alloc(xvcgpu_ctx *ctx, size_t bytes, void **host_ptr);
xvcgpu_status xvcgpu_host_free(xvcgpu_ctx *ctx, void *host_ptr);
xvcgpu_status xvcgpu_memcpy_h2d_async(xvcgpu_ctx *ctx, void *dst, const void *src,
                                      size_t bytes);
/* The read-back that does not wait: queued on the context's stream behind the
 * kernels that produce `src`; `dst` (page-locked, or the call may block) holds the
 * bytes once xvcgpu_sync / an event recorded after the call has completed.  Several
 * result arrays of one step are fetched with one wait. */
xvcgpu_status xvcgpu_memcpy_d2h_async(xvcgpu_ctx *ctx, void *dst, const void *src,
                                      size_t bytes);

/* ---- pictures ----------------------------------------------------------- *
 * Device twin of YuvPicture (yuv_pic.cc:32-68): planar Y,U,V of 16-bit
 * samples, every plane surrounded by a replicated border.  The reference uses
 * 80 luma / 40 chroma samples; here the border is 128 / 64 and strides are
 * rounded to 64 samples so that every row starts 256-byte aligned in HBM.
 * All MV clipping rules (inter_prediction.cc:769-782) keep reads within 80. */
#define XVCGPU_BORDER_LUMA 128
#define XVCGPU_BORDER_CHROMA 64

xvcgpu_status xvcgpu_picture_create(xvcgpu_ctx *ctx, int width, int height,
                                    int bitdepth, xvcgpu_picture **out);
/* Bytes needed by a picture of this size (for xvcgpu_picture_wrap). */
size_t xvcgpu_picture_bytes(int width, int height);
/* Build a picture over caller-owned device memory (e.g. a torch tensor) of at
 * least xvcgpu_picture_bytes() bytes, 256-byte aligned. */
xvcgpu_status xvcgpu_picture_wrap(xvcgpu_ctx *ctx, int width, int height,
                                  int bitdepth, void *dev_mem, size_t bytes,
                                  xvcgpu_picture **out);
void xvcgpu_picture_destroy(xvcgpu_picture *pic);
/* Host <-> device transfer of the visible area; planes are tightly described
 * by (pointer, stride in samples).  Cf. YuvPicture::CopyToSameBitdepth
 * (yuv_pic.cc:78-116) for the download direction. */
xvcgpu_status xvcgpu_picture_upload(xvcgpu_picture *pic,
                                    const uint16_t *const planes[3],
                                    const ptrdiff_t strides[3]);
xvcgpu_status xvcgpu_picture_download(const xvcgpu_picture *pic,
                                      uint16_t *const planes[3],
                                      const ptrdiff_t strides[3]);
/* Transfer including the border (tests of PadBorder). border_x/y <= 128/64
 * luma, halved for chroma. */
xvcgpu_status xvcgpu_picture_upload_padded(xvcgpu_picture *pic,
                                           const uint16_t *const planes[3],
                                           const ptrdiff_t strides[3],
                                           int border_luma);
xvcgpu_status xvcgpu_picture_download_padded(const xvcgpu_picture *pic,
                                             uint16_t *const planes[3],
                                             const ptrdiff_t strides[3],
                                             int border_luma);
/* Device pointer to sample (0,0) of a plane and its stride in samples. */
xvcgpu_status xvcgpu_picture_plane(const xvcgpu_picture *pic, int comp,
                                   void **dev_ptr, ptrdiff_t *stride);
xvcgpu_status xvcgpu_picture_copy(xvcgpu_ctx *ctx, xvcgpu_picture *dst,
                                  const xvcgpu_picture *src);

/* ---- P1: YuvPicture::PadBorder (yuv_pic.cc:118-150) --------------------- */
xvcgpu_status xvcgpu_pad_border(xvcgpu_ctx *ctx, xvcgpu_picture *pic);

/* ---- M1..M7: SampleMetric::Compare, Sample x Sample --------------------- *
 * (sample_metric.cc:171-223, fn table sample_metric.h:166-191).
 * One candidate = block (x,y,w,h) of component `comp` in pic `a` against the
 * block displaced by the full-pel vector (mv_x,mv_y) in pic `b`.
 * out[i] = static_cast<Distortion>(dist * weight), exactly as Compare(). */
typedef struct xvcgpu_metric_cand {
  int16_t x, y;       /* position in the component plane                    */
  uint8_t w, h;       /* size in the component plane                        */
  uint8_t metric;     /* xvcgpu_metric                                      */
  int8_t qp;          /* raw luma qp (structural SSD only)                  */
  int16_t mv_x, mv_y; /* full-pel displacement into `b`                     */
} xvcgpu_metric_cand;

xvcgpu_status xvcgpu_metric_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                  const xvcgpu_picture *b, int comp,
                                  double weight, int structural_strength,
                                  const xvcgpu_metric_cand *d_cands, int n,
                                  uint64_t *d_out);

/* ---- T4 building block: motion-compensate, then Compare ------------------- *
 * What InterSearch::GetSubpelDist (inter_search.cc:1022-1031), EvalStartMvp
 * (:966-997) and SearchMergeCandidates (:165-197) do per candidate MV:
 * MotionCompensationMv of the luma block (x,y,w,h) of `orig` from `ref` with
 * the 1/16-pel vector (mv_x,mv_y) (clipped inside, ClipMv), then
 * SampleMetric::CompareSample(orig, prediction) with `metric`.
 * out[i] = the Distortion (weight 1.0: luma). */
typedef struct xvcgpu_mc_metric_cand {
  int16_t x, y;         /* luma position                                     */
  uint8_t w, h;         /* luma size                                         */
  uint8_t metric;       /* xvcgpu_metric                                     */
  int8_t qp;            /* raw luma qp (structural SSD only)                 */
  int32_t mv_x, mv_y;   /* 1/16-pel vector into `ref`                        */
} xvcgpu_mc_metric_cand;

xvcgpu_status xvcgpu_mc_metric_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref,
                                     int structural_strength,
                                     const xvcgpu_mc_metric_cand *d_cands, int n,
                                     uint64_t *d_out);

/* ---- T1 + T3 (+W1, I1, M1, M4): MotionEstNormal with TZ search ---------- *
 * (inter_search.cc:606-662, inter_tz_search.cc:84-171, inter_search.cc:
 * 893-964).  One result per xvcgpu_me_block, bit-identical to running the
 * reference search on that block with the same predictor inputs.
 * Blocks: w, h in {4, 8, 16, 32, 64} (4x4 included: the reference's RD search
 * tries it); x, y multiples of 4
 * inside the picture.  The descriptors live in device memory, so a job the
 * search cannot take (any other size, or larger than max_block_size of
 * xvcgpu_me_search_sized) is reported in its result slot instead of an error
 * code: the XVCGPU_ME_UNSUPPORTED record - fullpel_cost = subpel_dist =
 * 0xffffffff, vectors 0 (no real result has cost 0xffffffff).
 * flags: XVCGPU_ME_FULLPEL runs the TZ search; XVCGPU_ME_SUBPEL runs the
 * 9+8 point half/quarter-pel refinement starting from results[i].fullpel_*
 * (taken from the TZ search when both flags are set).  XVCGPU_ME_LIC_JOBS: the
 * batch holds jobs with XVC_ME_USE_LIC (xvcgpu_types.h: AC-only metrics,
 * GetFullpelMetric / GetSubpelMetric inter_search.cc:1059-1076); their kernel
 * instances are launched beside the plain ones.  Without the flag such a job
 * is reported unsupported.  XVCGPU_ME_HINT_SQ16: a performance hint, never a
 * change of results - the caller expects (almost) every job of the 16 class to
 * be a 16x16 (or 16x8) CU, as in a picture's frame pass on a 16-sample CU grid:
 * the both-phases search then runs a kernel that takes only those shapes (five
 * waves per SIMD) and a second, small one that answers the jobs of any other size
 * a few per wave, one after the other - more slowly, same results.
 * XVCGPU_ME_ONLY_SQ16 (with both phases, max_block_size 16): the caller's word that
 * every job IS a 16x16 or 16x8 block - the second kernel is not launched, and a job
 * of any other shape is answered with the XVCGPU_ME_UNSUPPORTED record. */
#define XVCGPU_ME_FULLPEL 1
#define XVCGPU_ME_SUBPEL 2
#define XVCGPU_ME_LIC_JOBS 4
#define XVCGPU_ME_HINT_SQ16 8
#define XVCGPU_ME_ONLY_SQ16 16
#define XVCGPU_ME_UNSUPPORTED 0xffffffffu /* fullpel_cost / subpel_dist of a job not taken */
xvcgpu_status xvcgpu_me_search(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                               const xvcgpu_picture *ref, int flags,
                               const xvcgpu_me_block *d_blocks, int n,
                               xvcgpu_me_result *d_results);

/* Same, when the caller knows that no block of the batch exceeds
 * max_block_size (16, 32 or 64) in either dimension: only the kernel variants
 * whose LDS footprint is needed are launched. */
xvcgpu_status xvcgpu_me_search_sized(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref, int flags,
                                     const xvcgpu_me_block *d_blocks, int n,
                                     xvcgpu_me_result *d_results,
                                     int max_block_size);

/* ---- I1: MotionCompensationMv, uni-prediction --------------------------- *
 * (inter_prediction.cc:740-758 -> FilterLuma/FilterChroma :1387-1448).
 * Writes the predicted block of component blk.comp into `pred` at the CU's
 * position (pred plays the role of TransformEncoder's pred buffers,
 * transform_encoder.cc:45-47). */
xvcgpu_status xvcgpu_mc_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                              xvcgpu_picture *pred,
                              const xvcgpu_mc_block *d_blocks, int n);

/* I3 (LIC half): InterPrediction::MotionCompensationMv for uni-pred CUs that
 * use local illumination compensation (inter_prediction.cc:740-758,
 * LocalIlluminationComp :1555-1575, DeriveLicParams :1577-1663): the ordinary
 * prediction, then scale / offset from a linear model fitted to the block's
 * row above / column left: `rec` (the current picture's reconstruction, which
 * must hold the neighbouring CUs) against `ref` displaced by the rounded
 * full-pel vector.  Written into `pred` at the block's position.  The wave of
 * CUs of one call must not depend on each other's reconstruction. */
xvcgpu_status xvcgpu_mc_lic_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                  const xvcgpu_picture *rec, xvcgpu_picture *pred,
                                  const xvcgpu_mc_lic_block *d_blocks, int n);

/* ---- N1: the decoder's inter prediction, any inter CU ---------------------- *
 * InterPrediction::MotionCompensation (inter_prediction.cc:710-738) for n
 * (CU, component) jobs: uni- or bi-prediction from any of the picture's
 * reference pictures (refs[0..n_refs): the job names a slot per list),
 * translational or affine (MotionCompAffine :1044-1136, for either output
 * precision), with or without local illumination compensation (:1555-1663; the
 * model reads the neighbouring reconstruction from `rec`, so the CUs of one
 * call must not depend on each other).  Written into `pred` at the CU's
 * position.  n_refs <= 10 (two lists of kMaxNumRefPics, common.h:144). */
xvcgpu_status xvcgpu_inter_pred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *const *refs,
                                      int n_refs, const xvcgpu_picture *rec,
                                      xvcgpu_picture *pred,
                                      const xvcgpu_inter_block *d_blocks, int n);

/* The same with the RD search's destination: InterPrediction::MotionCompensation
 * writes into whatever SampleBuffer the caller passes (the encoder's
 * temp_pred_ / bipred buffers, inter_search.cc:96-99, :417), not into the
 * picture - overlapping candidate CUs of one CTU are then independent jobs.
 * Block i goes to `scratch` at d_dst[i] (luma position; chroma at half of it);
 * `scratch` is any picture of the same bit depth and chroma format, its size
 * is unrelated to the references'.  `rec` is read at the CUs' own positions
 * (LIC jobs only) and has the references' size. */
xvcgpu_status xvcgpu_inter_pred_batch_to(xvcgpu_ctx *ctx, const xvcgpu_picture *const *refs,
                                         int n_refs, const xvcgpu_picture *rec,
                                         xvcgpu_picture *scratch,
                                         const xvcgpu_inter_block *d_blocks,
                                         const xvcgpu_block_pos *d_dst, int n);

/* n block copies src -> dst (pictures of any two sizes): how the originals of
 * candidate CUs are brought beside their scratch predictions, so that the
 * residual / metric batches address one geometry (the reference passes
 * orig_pic_ + the CU position and a temp buffer side by side,
 * transform_encoder.cc:203-215).  Blocks must lie inside both pictures; the
 * destination blocks of one call must not overlap. */
xvcgpu_status xvcgpu_copy_blocks(xvcgpu_ctx *ctx, const xvcgpu_picture *src,
                                 xvcgpu_picture *dst, const xvcgpu_copy_block *d_blocks,
                                 int n);

/* The same for the blocks of a picture's CU loop in CU order - block 3 * cu + comp,
 * Y U V each, CUs up to 16x16 (what xvcgpu_fwd_from_me_classify leaves behind) -,
 * in place on `rec`, which holds the prediction: the U and V blocks of a CU share
 * a wave, blocks without levels cost a look at their count.  Same results as
 * xvcgpu_inv_transform_batch(ctx, rec, rec, d_blocks, 3 * n_cus, ...). */
xvcgpu_status xvcgpu_inv_transform_cu_order(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                            const xvcgpu_tx_block *d_blocks, int n_cus,
                                            const int16_t *d_levels,
                                            const uint32_t *d_level_offsets,
                                            const int32_t *d_nnz);

/* ---- I3 (affine half): MotionCompAffine -> Sample ------------------------- *
 * (inter_prediction.cc:1044-1136): the CU is cut into sub-blocks whose size
 * follows from the corner-MV differences, each sub-block gets its own MV
 * (8-bit-fraction interpolation of the corner MVs, clipped) and is
 * motion-compensated with the ordinary luma / chroma filters.  Chroma follows
 * the reference's C kernels (its SSE2 kernels differ on 2-wide blocks). */
xvcgpu_status xvcgpu_mc_affine_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                     xvcgpu_picture *pred,
                                     const xvcgpu_mc_affine_block *d_blocks, int n);

/* ---- I2: MotionCompensation of bi-predicted CUs --------------------------- *
 * (inter_prediction.cc:710-738: MotionCompUniPred -> int16 for both lists,
 * :1156-1172, then AddAvgBi :1545-1547).  ref0 / ref1 are the list-0 / list-1
 * reference pictures; writes component blk.comp of the CU into `pred`. */
xvcgpu_status xvcgpu_mc_bipred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref0,
                                     const xvcgpu_picture *ref1,
                                     xvcgpu_picture *pred,
                                     const xvcgpu_mc_bi_block *d_blocks, int n);

/* ---- M2 + T2 + T7: bi-prediction refinement search ------------------------ *
 * One inner step of InterSearch::SearchBiIterative (inter_search.cc:392-433)
 * per job: predict from the OTHER list with job.other_mv (ref_other), form
 * the int16 target 2*orig - pred (SubtractWeighted, sample_buffer.h:147-161),
 * then MotionEstNormal(kFullSearch, bipred=true, mv_bootstrap) on ref_search
 * (:606-662): FullSearch over DetermineMinMaxMv(boot, 4) with kSad/kSadFast
 * on the target (:853-891) and SubpelSearch with SATD on the target
 * (:893-964).  d_results[i]: fullpel_x/y = FullSearch result, mv_x/y = final
 * MV, subpel_dist = the function's *out_dist (SATD >> 1, :660),
 * fullpel_cost = 0.  max_block_size as in xvcgpu_me_search_sized; jobs that
 * cannot be taken get the XVCGPU_ME_UNSUPPORTED record. */
xvcgpu_status xvcgpu_bipred_search(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *ref_other,
                                   const xvcgpu_picture *ref_search,
                                   const xvcgpu_bi_block *d_jobs, int n,
                                   xvcgpu_me_result *d_results,
                                   int max_block_size);
/* The same step for CUs that try local illumination compensation
 * (cu.GetUseLic()): the other list's prediction is the compensated one
 * (inter_prediction.cc:710-722 -> LocalIlluminationComp :1555-1575; the model
 * reads the current reconstruction `rec` around the CU, d_neighbours[i] names
 * the CUs above / left of job i - only x, y of the neighbours and the
 * XVC_LIC_HAS_* bits of xvcgpu_mc_lic_block are read), the full-pel stage
 * compares with kSadAcOnly[Fast] and the sub-pel stage with kSatdAcOnly
 * (inter_search.cc:1059-1076), both on the int16 target. */
xvcgpu_status xvcgpu_bipred_search_lic(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                       const xvcgpu_picture *ref_other,
                                       const xvcgpu_picture *ref_search,
                                       const xvcgpu_picture *rec,
                                       const xvcgpu_bi_block *d_jobs,
                                       const xvcgpu_mc_lic_block *d_neighbours, int n,
                                       xvcgpu_me_result *d_results, int max_block_size);


/* Same for all three components of every CU of a motion search batch, taking
 * the MV from d_results[i].mv_* (InterPrediction::MotionCompensation for a
 * uni-pred CU, inter_prediction.cc:710-722): decisions stay in HBM. */
xvcgpu_status xvcgpu_mc_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                xvcgpu_picture *pred,
                                const xvcgpu_me_block *d_blocks,
                                const xvcgpu_me_result *d_results, int n);
/* Device-side host-driver glue: fill the deblocking metadata of n uni-pred
 * inter CUs (what CuEncoder stores into CodingUnit, cu_encoder.cc:543-577)
 * from the search results and the luma non-zero counts of the residual batch.
 * d_luma_tx_index[i] = index of CU i's luma block in d_nnz (NULL: identity). */
xvcgpu_status xvcgpu_cu_info_from_me(xvcgpu_ctx *ctx,
                                     const xvcgpu_me_block *d_blocks,
                                     const xvcgpu_me_result *d_results,
                                     const int32_t *d_nnz,
                                     const int32_t *d_luma_tx_index, int n,
                                     int qp_y, int qp_c, int ref_poc,
                                     xvcgpu_cu_info *d_cus);

/* ---- X1 + Q + Q1 + X2 + R1: TransformAndReconstruct --------------------- *
 * (transform_encoder.cc:203-285) with the reference's non-RDO quantiser
 * RdoQuant::QuantFast as shipped (rdo_quant.cc:156-201: sign-data hiding
 * CoeffSignHideFast :448-573 included; XVC_TXF_* bits in
 * xvcgpu_tx_block.intra_pic select the restricted mode / the coefficient scan
 * of small intra CUs).  For block i:
 * resi = orig - pred; coeff = T(resi); level = Q(coeff) -> d_levels +
 * d_level_offsets[i] (w*h int16, row-major, stride w); d_nnz[i] = non-zero
 * count; rec = cbf ? clip(pred + T^-1(Q^-1(level))) : pred.
 * d_levels / d_level_offsets / d_nnz may be NULL when not wanted. */
xvcgpu_status xvcgpu_residual_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                    const xvcgpu_picture *pred,
                                    xvcgpu_picture *rec,
                                    const xvcgpu_tx_block *d_blocks, int n,
                                    int16_t *d_levels,
                                    const uint32_t *d_level_offsets,
                                    int32_t *d_nnz);
/* The same with the quantiser the reference's encoder always runs (Q2:
 * RdoQuant::QuantRdo with CoeffSignHideRdo, rdo_quant.cc:203-446, :575-687;
 * selected by transform_encoder.cc:230 because encoder_settings.h:59 is
 * rdo_quant = true) for the blocks whose intra_pic field carries XVC_TXF_RDOQ;
 * the others take QuantFast as above.  d_params[i] belongs to d_blocks[i]
 * (lambda and rd_factor in fixed point, computed by the host from the
 * reference's doubles; which context snapshot; intra / inter CU);
 * d_contexts[d_params[i].ctx_index] = the CABAC context states of the syntax
 * writer the reference would pass (rdo_quant.cc:254), read through
 * ContextModel::GetEntropyBits only.  Levels, non-zero counts and the
 * reconstruction are bit-identical to the reference's for any block size
 * 2..64 x 2..64 and any transform type pair. */
xvcgpu_status xvcgpu_residual_rdoq_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                         const xvcgpu_picture *pred, xvcgpu_picture *rec,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         int16_t *d_levels, const uint32_t *d_level_offsets,
                                         int32_t *d_nnz,
                                         const xvcgpu_rdoq_contexts *d_contexts,
                                         const xvcgpu_rdoq_params *d_params);
/* The same for the blocks of ONE CU state (n <= 64) whose original and prediction lie
 * elsewhere: block i reads its original at d_src_pos[2 i] of `orig` and its prediction
 * at d_src_pos[2 i + 1] of `pred` (positions in the plane of the block's component) and
 * writes its reconstruction at its own (x, y) of `rec` - the RD search's alternatives
 * of a CU (CompressAndEvalTransform's default / transform-select / skip candidates)
 * each reconstruct into their own slot of a scratch picture from the one prediction
 * and the original picture itself, without the two block copies that put them side by
 * side.  orig may have another size than pred / rec.
 * d_eval_cands != NULL: the evaluation's distortions in the same launch
 * (xvcgpu_eval_dist_batch's arithmetic, xvcgpu_eval_cand): candidates [0, n_eval_head)
 * - the prediction against the original, the cbf-zero distortions - and candidate
 * n_eval_head + i, priced by the workgroup that has just reconstructed block i;
 * d_eval_out[c] for every candidate.  With it a CompressAndEvalCbf is two launches:
 * the prediction, and this. */
xvcgpu_status xvcgpu_residual_rdoq_batch_at(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                            const xvcgpu_picture *pred, xvcgpu_picture *rec,
                                            const xvcgpu_tx_block *d_blocks, int n,
                                            int16_t *d_levels, const uint32_t *d_level_offsets,
                                            int32_t *d_nnz,
                                            const xvcgpu_rdoq_contexts *d_contexts,
                                            const xvcgpu_rdoq_params *d_params,
                                            const xvcgpu_block_pos *d_src_pos,
                                            int structural_strength,
                                            const xvcgpu_eval_cand *d_eval_cands, int n_eval_head,
                                            uint64_t *d_eval_out);


/* Q2 alone: RdoQuant::QuantRdo on coefficients the caller holds (the output
 * of xvcgpu_fwd_transform_batch; the levels go to xvcgpu_inv_transform_batch):
 * block i reads w*h int16 at d_coeffs + d_offsets[i] and writes its levels at
 * d_levels + d_offsets[i], d_nnz[i] = RdoQuant::QuantRdo's return value (all
 * levels zero when it is 0).  n_coeffs = the number of int16 in d_coeffs
 * (sizes the per-coefficient scratch).  Only w, h, comp, qp and the
 * XVC_TXF_NO_SIGN_HIDING / scan bits of a block are read.  This is the
 * throughput form: several blocks share a wave (16 blocks up to 8x8, 4 up to
 * 16x16).  Blocks whose XVC_RDOQ_NO_2X2 case applies (2-wide, rdo_quant_2x2
 * off) are NOT taken here: route them to xvcgpu_residual_batch. */
xvcgpu_status xvcgpu_quant_rdo_batch(xvcgpu_ctx *ctx, int bitdepth,
                                     const xvcgpu_tx_block *d_blocks, int n,
                                     const int16_t *d_coeffs, const uint32_t *d_offsets,
                                     size_t n_coeffs, int16_t *d_levels, int32_t *d_nnz,
                                     const xvcgpu_rdoq_contexts *d_contexts,
                                     const xvcgpu_rdoq_params *d_params);

/* The two calls above with the classification pass of the quantiser (which
 * blocks hold a coefficient that quantises to a level at all - at QP 32 one in
 * eight) done by the forward transform, on the coefficients it has at hand:
 * xvcgpu_fwd_from_me_classify = xvcgpu_fwd_from_me + the blocks' classes into
 * the context's scratch, zero levels and d_nnz = 0 for the blocks the quantiser
 * has nothing to do for (their coefficients are not stored);
 * xvcgpu_quant_rdo_classified_batch = xvcgpu_quant_rdo_batch over the SAME
 * blocks (3 per CU: Y, U, V, in CU order) without its own pass over all
 * coefficients.  Nothing else of the context's RDOQ entry points may run in
 * between.  Same results as the plain pair.  d_cus (optional, both calls): the
 * CUs' deblocking records (xvcgpu_cu_info_from_me) written on the way - the
 * forward call writes them with cbf_luma = 0, the quantiser sets the flag of
 * the CUs whose luma block keeps a level. */
xvcgpu_status xvcgpu_fwd_from_me_classify(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *ref, xvcgpu_picture *pred,
                                          const xvcgpu_me_block *d_blocks,
                                          const xvcgpu_me_result *d_results, int n, int qp_y,
                                          int qp_c, int ref_poc, int16_t *d_coeffs,
                                          const uint32_t *d_coeff_offsets, size_t n_coeffs,
                                          int16_t *d_levels, int32_t *d_nnz,
                                          xvcgpu_cu_info *d_cus);
/* ... and with the quantiser's context snapshots and per-block parameters (the
 * ones xvcgpu_quant_rdo_classified_batch will be given, blocks 3 * cu + comp):
 * the forward transform then also runs the all-zero proof
 * (xvcgpu_quant_rdo_set_prove_zero, below) on the coefficients it holds - no
 * launch of its own, no second read - unless the proof is switched off (mode 0).
 * Both null: xvcgpu_fwd_from_me_classify. */
xvcgpu_status xvcgpu_fwd_from_me_classify_prove(
    xvcgpu_ctx *ctx, const xvcgpu_picture *orig, const xvcgpu_picture *ref, xvcgpu_picture *pred,
    const xvcgpu_me_block *d_blocks, const xvcgpu_me_result *d_results, int n, int qp_y, int qp_c,
    int ref_poc, int16_t *d_coeffs, const uint32_t *d_coeff_offsets, size_t n_coeffs,
    int16_t *d_levels, int32_t *d_nnz, xvcgpu_cu_info *d_cus,
    const xvcgpu_rdoq_contexts *d_contexts, const xvcgpu_rdoq_params *d_params);
xvcgpu_status xvcgpu_quant_rdo_classified_batch(xvcgpu_ctx *ctx, int bitdepth,
                                                const xvcgpu_tx_block *d_blocks, int n,
                                                const int16_t *d_coeffs,
                                                const uint32_t *d_offsets, size_t n_coeffs,
                                                int16_t *d_levels, int32_t *d_nnz,
                                                const xvcgpu_rdoq_contexts *d_contexts,
                                                const xvcgpu_rdoq_params *d_params,
                                                xvcgpu_cu_info *d_cus);

/* Sizes the context's scratch for batches of up to n blocks / n_coeffs
 * coefficients now, so that later xvcgpu_quant_rdo_batch calls never allocate
 * (required before recording them, xvcgpu_record_begin). */
xvcgpu_status xvcgpu_quant_rdo_reserve(xvcgpu_ctx *ctx, int n, size_t n_coeffs);

/* The all-zero proof ahead of the quantiser's walk (k_rdoq.h, rdoq_prove_zero_kernel):
 * blocks for which RdoQuant::QuantRdo (rdo_quant.cc:223-446) is bound to return 0 -
 * provable from the plain quantised magnitudes and the context snapshot - get their
 * zero levels from one more launch and never reach the walk.  Results are the same
 * either way; the launch pays when the walk is throughput bound (large batches, few
 * surviving levels) and costs a few percent when it is latency bound.
 * mode: 0 never, 1 always, -1 (default) for batches of XVCGPU_PROVE_ZERO_AUTO_BLOCKS
 * blocks or more - and, where the batch follows xvcgpu_fwd_from_me_classify, which
 * knows the picture's QP, only from XVCGPU_PROVE_ZERO_AUTO_QP up (measured with the
 * proof as its own launch, frame passes/s without -> with: 3840x2160 QP 27 1475 ->
 * 1440, QP 32 1841 -> 2036; 7680x4320 QP 37 525 -> 551; 1920x1080 QP 32 6811 ->
 * 6696).  A batch that follows xvcgpu_fwd_from_me_classify_prove needs no launch:
 * that call has run the proof already, for any mode but 0.  The environment
 * variable XVCGPU_PROVE_ZERO (0 / 1), read by xvcgpu_create, sets the initial mode. */
#define XVCGPU_PROVE_ZERO_AUTO_BLOCKS 65536
#define XVCGPU_PROVE_ZERO_AUTO_QP 30
xvcgpu_status xvcgpu_quant_rdo_set_prove_zero(xvcgpu_ctx *ctx, int mode);

/* Diagnostics: the number of blocks of the last xvcgpu_quant_rdo_batch that
 * needed the walk, by class (out[0]: diagonal scan, up to four 4x4 sub-blocks;
 * out[1]: diagonal scan, up to sixteen, sides up to 32 - both walked with four
 * lanes per sub-block; out[2]: everything else - more sub-blocks, 64-point sides,
 * the horizontal / vertical scans, 2-wide blocks); blocks whose coefficients all
 * quantise to zero are settled by the classification pass and not counted.
 * Synchronises the stream. */
xvcgpu_status xvcgpu_quant_rdo_class_counts(xvcgpu_ctx *ctx, int32_t out[3]);
/* The walk is two launches: the four-lane classes (out[0], out[1] above) and the
 * general class (out[2]), whose waves hold 255 vector registers - even with an empty
 * list that launch waits for room beside other streams' kernels (190 us in flight
 * per 2160p picture with three picture chains).  A caller whose batches only hold
 * diagonal-scan blocks with 4x4 sub-blocks, sides up to 32 and at most sixteen
 * sub-blocks (a frame pass of CUs up to 16x16, say) sets on = 1 and the general
 * launch is not made.  A block outside that promise is not quantised; the next
 * xvcgpu_sync reports it as XVCGPU_INVALID_ARGUMENT (xvcgpu_last_error says why). */
xvcgpu_status xvcgpu_quant_rdo_set_four_lane_only(xvcgpu_ctx *ctx, int on);

/* I1 + the above fused, for the uni-pred inter CUs of a motion search batch
 * (InterSearch::CompressAndEvalCbf without the RD bookkeeping,
 * inter_search.cc:261-365): for CU i and each component, motion-compensate
 * with d_results[i].mv_*, run TransformAndReconstruct (DCT-2, QuantFast) into
 * `rec`, store the non-zero count at d_nnz[3*i + comp] and - when d_cus is
 * given - CU i's deblocking metadata at d_cus[i].  The prediction stays in
 * LDS.  CUs up to 16x16 with both dimensions >= 8 (else use xvcgpu_mc_from_me
 * + xvcgpu_residual_batch + xvcgpu_cu_info_from_me, which compute the same). */
xvcgpu_status xvcgpu_recon_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *ref, xvcgpu_picture *rec,
                                   const xvcgpu_me_block *d_blocks,
                                   const xvcgpu_me_result *d_results, int n,
                                   int qp_y, int qp_c, int intra_pic, int ref_poc,
                                   int32_t *d_nnz, xvcgpu_cu_info *d_cus);

/* The front half of that pipeline alone, for a quantiser that runs as its own
 * kernel (xvcgpu_quant_rdo_batch): MotionCompensationMv of CU i into `pred`
 * (TransformEncoder's prediction buffer) and the forward transform of orig -
 * pred, coefficients of the CU's Y / U / V block at d_coeffs +
 * d_coeff_offsets[3 * i + comp] (as xvcgpu_fwd_transform_batch writes them).
 * CUs up to 16x16. */
xvcgpu_status xvcgpu_fwd_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                 const xvcgpu_picture *ref, xvcgpu_picture *pred,
                                 const xvcgpu_me_block *d_blocks,
                                 const xvcgpu_me_result *d_results, int n, int16_t *d_coeffs,
                                 const uint32_t *d_coeff_offsets);
/* The same with RdoQuant::QuantRdo (see xvcgpu_residual_rdoq_batch):
 * d_params[3 * i + comp] belongs to component comp of CU i; tx_flags = the
 * XVC_TXF_* bits common to all blocks (XVC_TXF_RDOQ is implied). */
xvcgpu_status xvcgpu_recon_from_me_rdoq(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                        const xvcgpu_picture *ref, xvcgpu_picture *rec,
                                        const xvcgpu_me_block *d_blocks,
                                        const xvcgpu_me_result *d_results, int n,
                                        int qp_y, int qp_c, int tx_flags, int ref_poc,
                                        int32_t *d_nnz, xvcgpu_cu_info *d_cus,
                                        const xvcgpu_rdoq_contexts *d_contexts,
                                        const xvcgpu_rdoq_params *d_params);

/* X1 only: coefficients of (orig - pred) for a host-side quantiser (RDOQ stays
 * on the host). Output layout as d_levels. */
xvcgpu_status xvcgpu_fwd_transform_batch(xvcgpu_ctx *ctx,
                                         const xvcgpu_picture *orig,
                                         const xvcgpu_picture *pred,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         int16_t *d_coeffs,
                                         const uint32_t *d_coeff_offsets);
/* Q1 + X2 + R1: Quantize::Inverse -> InverseTransform -> AddClip for levels
 * chosen by the host (decoder reconstruction shares this entry point,
 * cu_decoder.cc:102-138). d_nnz[i] == 0 copies pred; dc-only shortcut applied
 * when d_nnz[i] == 1 and level[0] != 0 (transform_encoder.cc:241). 
 * `pred` and `rec` may be the same picture (the prediction was written where
 * the reconstruction goes): the blocks without levels then cost nothing. */
xvcgpu_status xvcgpu_inv_transform_batch(xvcgpu_ctx *ctx,
                                         const xvcgpu_picture *pred,
                                         xvcgpu_picture *rec,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         const int16_t *d_levels,
                                         const uint32_t *d_level_offsets,
                                         const int32_t *d_nnz);
/* The same, and the RD loop's distortion of the coded block in the residual
 * domain (M6 short-short: SampleMetric::CompareShort on temp_resi_orig_ /
 * temp_resi_, transform_encoder.cc:69-77, sample_metric.cc:286-290):
 * d_dist[i] = sum((orig - pred - T^-1(Q^-1(level)))^2) >> 2 (bitdepth - 8) over
 * block i - what TransformAndReconstruct's caller prices an inter block with
 * (fast_inter_transform_dist); the chroma distortion weight (:275-276) is the
 * caller's.  For a block without levels the reconstructed residual is zero. */
xvcgpu_status xvcgpu_inv_transform_dist_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                              const xvcgpu_picture *pred, xvcgpu_picture *rec,
                                              const xvcgpu_tx_block *d_blocks, int n,
                                              const int16_t *d_levels,
                                              const uint32_t *d_level_offsets,
                                              const int32_t *d_nnz, uint64_t *d_dist);

/* ---- C1, decision half ----------------------------------------------------- *
 * TransformEncoder::CompressAndEvalTransform's choice among a component's
 * alternatives (transform_encoder.cc:53-201: default transform, all-zero block,
 * transform skip, the transform-select indices) and the tail of
 * InterSearch::CompressAndEvalCbf (inter_search.cc:316-361: root-cbf-zero test,
 * gate of the transform-select second pass), batched.  The distortions are the
 * outputs of xvcgpu_residual_rdoq_batch + xvcgpu_metric_batch /
 * xvcgpu_inv_transform_dist_batch (device arrays the caller gathers into the
 * alternative records, or fills from the host); the bits are what the caller's
 * RdoSyntaxWriter prices each alternative with.  Costs are formed and compared
 * exactly as the reference does: dist + (Cost)(bits * lambda + 0.5), first
 * strictly cheaper alternative wins.  One thread per job. */
xvcgpu_status xvcgpu_tx_eval_batch(xvcgpu_ctx *ctx, const xvcgpu_tx_eval_job *d_jobs, int n,
                                   const xvcgpu_tx_eval_alt *d_alts,
                                   xvcgpu_tx_eval_result *d_out);
xvcgpu_status xvcgpu_root_cbf_batch(xvcgpu_ctx *ctx, const xvcgpu_root_cbf_job *d_jobs, int n,
                                    xvcgpu_root_cbf_result *d_out);

/* ---- D1..D4: DeblockingFilter::DeblockPicture --------------------------- *
 * (deblocking_filter.cc:56-77). d_cu_map: one int32 per 4x4 luma cell,
 * row-major, `map_stride` entries per row, ceil(height/4) rows, value = index
 * into d_cus or -1. subblock_size 4 (default) or 8 (restricted mode). */
xvcgpu_status xvcgpu_deblock(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                             const xvcgpu_cu_info *d_cus, int n_cus,
                             const int32_t *d_cu_map, int map_stride,
                             int pic_is_bipred, int beta_offset, int tc_offset,
                             int subblock_size);

/* One pass (0 = vertical edges, 1 = horizontal edges) restricted to the
 * subblock rows y in [y_begin, y_end): the unit of CTU-row sharding.  Pass 1
 * of a shard must include the first subblock row of the shard below
 * (y_end = shard_end + subblock_size) once that shard's pass-0 rows have been
 * received (scheme B).  xvcgpu_deblock() == pass 0 then
 * pass 1 over [0, height). */
xvcgpu_status xvcgpu_deblock_rows(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                  const xvcgpu_cu_info *d_cus, int n_cus,
                                  const int32_t *d_cu_map, int map_stride,
                                  int pic_is_bipred, int beta_offset,
                                  int tc_offset, int subblock_size, int pass,
                                  int y_begin, int y_end);

/* One CU tree's share of DeblockPicture (deblocking_filter.cc:56-77): both
 * passes over the whole picture, filtering only the planes in comp_mask (1 =
 * luma, 2 = chroma, 3 = both).  An intra picture codes luma and chroma in two
 * CU trees (picture_data.cc:71-76): the primary tree filters luma on the
 * 4-sample grid, the secondary tree chroma on the 8-sample grid (:63-75).
 * xvcgpu_deblock() == comp_mask 3. */
xvcgpu_status xvcgpu_deblock_tree(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                  const xvcgpu_cu_info *d_cus, int n_cus,
                                  const int32_t *d_cu_map, int map_stride,
                                  int pic_is_bipred, int beta_offset, int tc_offset,
                                  int subblock_size, int comp_mask);

/* The tail of PictureEncoder::Encode in one pass over the picture
 * (picture_encoder.cc:141-151; one launch + the fold of the SSD parts):
 * DeblockPicture (deblocking_filter.cc:56-77, both passes, luma and chroma, the
 * default 4-sample subblocks) of `src` -> `dst`, PadBorder of `dst`
 * (yuv_pic.cc:118-150) and, when `orig` is given, the luma ComparePicture parts
 * of xvcgpu_picture_ssd(orig, dst, 0, shift_bitdepth) in d_ssd[0..1].
 * `src` (the unfiltered reconstruction; its border is not read for the result)
 * and `dst` must be different pictures of the same size.  Only for pictures
 * whose CUs are all at least 8x8 (every coding edge on the 8-sample grid, so no
 * two edges interact): the caller knows its CU tree; other pictures take
 * xvcgpu_deblock + xvcgpu_pad_border + xvcgpu_picture_ssd.  Width and height
 * must be multiples of 8 (XVCGPU_INVALID_ARGUMENT otherwise). */
xvcgpu_status xvcgpu_deblock_pad_ssd(xvcgpu_ctx *ctx, const xvcgpu_picture *src,
                                     xvcgpu_picture *dst, const xvcgpu_picture *orig,
                                     const xvcgpu_cu_info *d_cus, int n_cus,
                                     const int32_t *d_cu_map, int map_stride,
                                     int pic_is_bipred, int beta_offset, int tc_offset,
                                     int shift_bitdepth, uint64_t *d_ssd);

/* ---- picture SSD / PSNR parts ------------------------------------------- *
 * SampleMetric::ComparePicture / ComputePsnr block walk (sample_metric.cc:
 * 37-155): per-64x64-block SSD, each >> 2*(shift_bitdepth-8), summed.
 * d_out[0] = distortion, d_out[1] = sample count. */
xvcgpu_status xvcgpu_picture_ssd(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                 const xvcgpu_picture *b, int comp,
                                 int shift_bitdepth, uint64_t *d_out);
/* The share of the blocks whose first row (in the component plane) lies in
 * [y_begin, y_end): shares of disjoint row ranges covering the picture add up
 * to xvcgpu_picture_ssd - what one CU-row shard contributes. */
xvcgpu_status xvcgpu_picture_ssd_rows(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                      const xvcgpu_picture *b, int comp,
                                      int shift_bitdepth, int y_begin, int y_end,
                                      uint64_t *d_out);

/* ---- whole-picture passes around the hot path -------------------------- *
 * Decision-free, HBM-bound passes the encoder / decoder run on whole pictures
 * before and after the per-CU work; on the device so that planes never have to
 * visit the host.  All on the context's stream.
 *
 * Resampler::ConvertFrom without resizing (resample.cc:32-63, :152-262): packed
 * planar 4:2:0 input in device memory (Y, U, V back to back, rows tightly
 * packed; 1 byte per sample when in_bitdepth == 8, else 2 little endian) ->
 * the picture at its internal bit depth (samples << (bd - in_bitdepth)); when
 * the picture is larger than the input (internal size rounded up to the
 * minimum CU size) the last column / row is repeated.  Visible area only: call
 * xvcgpu_pad_border afterwards if the picture is searched. */
xvcgpu_status xvcgpu_picture_import(xvcgpu_ctx *ctx, xvcgpu_picture *pic,
                                    const void *d_src, int in_width, int in_height,
                                    int in_bitdepth);
/* Resampler::ConvertTo without resizing (resample.cc:96-147,
 * CopyToBytesWithShift :304-338 with the sample functions :475-551): the
 * display_width x display_height top-left part of the picture -> packed planar
 * bytes at out_bitdepth (same layout as above): copy / left shift when
 * out_bitdepth >= bd, else the rounding down-shift or, with `dither`, the
 * reference's error-feedback down-shift whose remainder runs through all
 * samples of a plane in raster order (computed here as a parallel prefix
 * sum). */
xvcgpu_status xvcgpu_picture_export(xvcgpu_ctx *ctx, const xvcgpu_picture *pic,
                                    void *d_dst, int display_width, int display_height,
                                    int out_bitdepth, int dither);
/* Checksum::CalculateCrc (checksum.cc:46-92): CRC-16 (0x1021, preset 0xffff,
 * low byte of a sample first, 16 trailing zero bits).  mode 0 = kMinOverhead:
 * one value over Y,U,V, d_hash[0..1]; mode 1 = kMaxRobust: one per plane,
 * d_hash[0..5] (high byte first).  d_hash: 8 bytes of device memory.  (The
 * reference's default, MD5, is a serial chain and stays on the host.) */
xvcgpu_status xvcgpu_picture_crc(xvcgpu_ctx *ctx, const xvcgpu_picture *pic, int mode,
                                 uint8_t *d_hash);
/* CuEncoder::CalcDeltaQpFromVariance (cu_encoder.cc:308-357), the integer
 * part: d_var16[by * ceil(w/16) + bx] = calc_variance of the 16x16 luma block
 * (blocks that hang over the picture edge read the border); and, if d_ctu_var
 * is not NULL, per ctu_size x ctu_size CTU in raster order the statistic the
 * QP offset is computed from: 1 + sorted in-picture block variances[blocks/2].
 * The host applies strength * (1.5 * log(v) - 15 - 2 * (bd - 8)). */
xvcgpu_status xvcgpu_variance_map(xvcgpu_ctx *ctx, const xvcgpu_picture *pic,
                                  uint64_t *d_var16, int ctu_size, uint64_t *d_ctu_var);
/* PictureEncoder::DetermineAllowLic (picture_encoder.cc:230-281): *d_out = sum
 * over the sample values of |histogram(luma of a) - histogram(luma of b)|; the
 * caller compares with (int)(0.06 * w * h). */
xvcgpu_status xvcgpu_histogram_distance(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                        const xvcgpu_picture *b, int64_t *d_out);

/* ---- intra prediction and SATD mode pre-selection ----------------------- *
 * IntraPrediction::FillReferenceState + Predict (intra_prediction.cc:81-147;
 * ComputeRefSamples :707-848, FilterRefSamples :850-871, planar / DC / angular
 * with the edge filters :365-558), 67-mode set with the default restriction
 * flags, for a batch of independent blocks: each job reads its reference
 * samples from `rec` (the neighbours it declares available must already be
 * reconstructed there) and writes the prediction of job.mode into `pred` at
 * the block's position.  Any component; comp 0 applies the luma rules
 * (filtered references, edge filters up to 16x16).  Chroma jobs may also ask for
 * mode XVC_INTRA_MODE_LM_CHROMA (PredLmChroma, :560-686, :873-906): the linear
 * model from the CU's own reconstructed luma, which must be in `rec` already
 * (availability = picture edges only, as in the reference). */
xvcgpu_status xvcgpu_intra_pred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *rec,
                                      xvcgpu_picture *pred,
                                      const xvcgpu_intra_block *d_jobs, int n);
/* The decoder's intra picture in ONE launch: all dependency waves of
 * CuDecoder::DecompressCu for intra CUs (cu_decoder.cc:100-166: Predict,
 * InverseTransform, AddClip).  Wave w = jobs d_wave_first[w] ..
 * d_wave_first[w + 1] of BOTH lists - d_jobs[k] (the prediction, any size / mode,
 * LM chroma included) and d_blocks[k] (the same block's transform job) - whose
 * reference samples come from waves < w; levels / offsets / nnz as for
 * xvcgpu_inv_transform_batch.  A cooperative launch (the workgroups meet at a
 * grid barrier between waves): XVCGPU_UNSUPPORTED when the device refuses it -
 * the caller then issues xvcgpu_intra_pred_batch + xvcgpu_inv_transform_batch
 * per wave, the same result. */
xvcgpu_status xvcgpu_intra_recon_waves(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                       xvcgpu_picture *pred,
                                       const xvcgpu_intra_block *d_jobs,
                                       const xvcgpu_tx_block *d_blocks,
                                       const int32_t *d_wave_first, int n_waves,
                                       const int16_t *d_levels,
                                       const uint32_t *d_level_offsets,
                                       const int32_t *d_nnz);
/* The prediction + SATD loop of IntraSearch::DetermineSlowIntraModes
 * (intra_search.cc:189-305) for luma blocks, all 67 modes per job (the reference
 * evaluates the even modes, then the odd neighbours of the best ones: any
 * subset it asks for is in the table): d_dist[job * 67 + mode] =
 * SampleMetric(kSatd)::CompareSample(original block, prediction).  The host
 * adds bits * lambda_sqrt from its entropy coder state and sorts.
 * max_block_size: an upper bound (4..64) of the block sides in the batch; it
 * sizes the on-chip tiles (batches of small blocks run at higher occupancy).
 * A job with a larger side is answered with d_dist[job][mode] = 0xffffffff for
 * every mode (XVCGPU_INTRA_SATD_UNSUPPORTED). */
#define XVCGPU_INTRA_SATD_UNSUPPORTED 0xffffffffu
xvcgpu_status xvcgpu_intra_satd_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                      const xvcgpu_picture *rec,
                                      const xvcgpu_intra_block *d_jobs, int n,
                                      uint32_t *d_dist, int max_block_size);

/* The fold of the fast pass kept on the device, for pipelines that do not want a
 * host round trip per batch: mode[cu] = first arg min over the 67 modes of
 * d_dist[cu][mode] (+ d_mode_cost[cu][mode] when not NULL: the caller's rate
 * term, e.g. bits * lambda_sqrt rounded).  Written to d_modes[cu] (may be NULL)
 * and, when given, into the `per_cu` (1..3) consecutive prediction jobs of the
 * CU (every component takes the luma mode, i.e. DM chroma) and into the scan
 * bits (XVC_TXF_SCAN_SHIFT) of its `per_cu` consecutive transform blocks, luma
 * block first, by TransformHelper::DetermineScanOrder (transform.cc:1614-1637).
 * The reference's own decision (stable sort by double cost, N best to the slow
 * RDO pass, intra_search.cc:236-303) remains host work on d_dist. */
xvcgpu_status xvcgpu_intra_select_modes(xvcgpu_ctx *ctx, const uint32_t *d_dist,
                                        const uint32_t *d_mode_cost, int n,
                                        int32_t *d_modes, xvcgpu_intra_block *d_jobs,
                                        xvcgpu_tx_block *d_blocks, int per_cu);

/* Predict + TransformAndReconstruct in one launch for intra blocks up to 16x16
 * (IntraSearch::PredictAndTransform, intra_search.cc:172-187, with QuantFast;
 * the decoder's CuDecoder::DecompressIntra when `orig` is NULL): job i =
 * d_jobs[i] (the prediction) + d_blocks[i] (its transform block: same
 * component, position and size).  The prediction never leaves the chip.
 * orig != NULL: levels / nnz are written as by xvcgpu_residual_batch (the three
 * arrays may be NULL); orig == NULL: they are read as by
 * xvcgpu_inv_transform_batch.  In place on `rec`; the jobs of one call must not
 * depend on each other.  Larger blocks, 2-wide blocks, the 4x4 DST and
 * transform skip are not taken (use xvcgpu_intra_pred_batch +
 * xvcgpu_residual_batch / xvcgpu_inv_transform_batch). */
xvcgpu_status xvcgpu_intra_recon_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                       xvcgpu_picture *rec,
                                       const xvcgpu_intra_block *d_jobs,
                                       const xvcgpu_tx_block *d_blocks, int n,
                                       int16_t *d_levels, const uint32_t *d_level_offsets,
                                       int32_t *d_nnz);

/* ---- T5: affine motion estimation ------------------------------------------ *
 * InterSearch::MotionEstAffine (inter_search.cc:664-749) for n (CU, list,
 * ref_idx) jobs on one reference picture - uni-pred (orig is the original
 * picture, 7 iterations at most) or, XVC_AFFINE_ME_BIPRED, the refinement
 * search of SearchBiIterative (:394-435) against 2 * orig - the other list's
 * affine prediction from `ref_other` (may be NULL when no job has the flag):
 * prediction from the affine predictor (and the optional bootstrap vector),
 * then per iteration AffineGradientSearch (:751-851: Sobel gradients, the 4x5
 * normal equations, elimination with partial pivoting, lround), DeriveMvAffine
 * (inter_prediction.cc:615-630), MotionCompAffine and the SATD / mvd-bit cost.
 * The reference's float / double arithmetic is reproduced exactly (every sum of
 * the normal equations is an exact multiple of 1/64 below 2^53, so its order
 * does not matter; the elimination is the same IEEE double sequence).
 * w, h in {16, 32, 64} (CodingUnit::CanUseAffine); any other shape is answered
 * with dist = iterations = 0xffffffff (XVCGPU_AFFINE_ME_UNSUPPORTED). */
#define XVCGPU_AFFINE_ME_UNSUPPORTED 0xffffffffu
xvcgpu_status xvcgpu_affine_me_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref,
                                     const xvcgpu_picture *ref_other,
                                     const xvcgpu_affine_me_block *d_blocks, int n,
                                     xvcgpu_affine_me_result *d_results);

/* ---- one step of a CU state's SearchMotion into all its reference pictures -------- *
 * InterSearch::SearchMotion (inter_search.cc:199-259) runs every step once per (list,
 * reference picture) of the CU: SearchRefIdx's EvalStartMvp and MotionEstNormal
 * (:536-578), SearchBiIterative's refinement per pair of pictures (:392-433), the affine
 * searches (:664-749).  With the single-picture entry points above that is one launch per
 * picture with one job each, one after the other.  The *_refs forms take the CU's
 * reference pictures as a table (refs[0 .. n_refs), n_refs <= 10 = 2 lists x
 * kMaxNumRefPics) and per job the slot(s) of the picture(s) it works on, so one step is
 * ONE launch whose jobs run side by side.  d_slots: one byte per job
 * (xvcgpu_me_search_refs, xvcgpu_mc_metric_batch_refs) or two (the searched picture, then
 * the other list's: xvcgpu_bipred_search_refs, xvcgpu_affine_me_batch_refs; a uni-pred
 * affine job may name 255 as its second); a first slot >= n_refs marks "no job": its
 * result is left as it was.
 * The jobs of a CU state all have the CU's size: block_class (16, 32 or 64 = the class of
 * max(w, h) as xvcgpu_me_search_sized forms it) / cu_height name the one set of kernel
 * instances that is launched; a job of another class is not searched (16: it is
 * answered XVCGPU_ME_UNSUPPORTED).  Results are those of the single-picture calls, job
 * for job.  No LIC jobs (XVCGPU_ME_LIC_JOBS is refused). */
xvcgpu_status xvcgpu_me_search_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                    const xvcgpu_picture *const *refs, int n_refs, int flags,
                                    const xvcgpu_me_block *d_blocks, const uint8_t *d_slots,
                                    int n, xvcgpu_me_result *d_results, int block_class);
xvcgpu_status xvcgpu_bipred_search_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                        const xvcgpu_picture *const *refs, int n_refs,
                                        const xvcgpu_bi_block *d_jobs, const uint8_t *d_slots,
                                        int n, xvcgpu_me_result *d_results, int block_class);
xvcgpu_status xvcgpu_mc_metric_batch_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *const *refs, int n_refs,
                                          int structural_strength,
                                          const xvcgpu_mc_metric_cand *d_cands,
                                          const uint8_t *d_slots, int n, uint64_t *d_out);
xvcgpu_status xvcgpu_affine_me_batch_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *const *refs, int n_refs,
                                          const xvcgpu_affine_me_block *d_blocks,
                                          const uint8_t *d_slots, int n,
                                          xvcgpu_affine_me_result *d_results, int cu_height);

/* ---- the folds of one SearchMotion chain --------------------------------------- *
 * A pass record - xvcgpu_cs_pass, include/xvcgpu_types.h - describes one SearchMotion of a CU; the three
 * calls below run the host logic of InterSearch::SearchRefIdx / SearchBiIterative /
 * SearchMotion (inter_search.cc:199-259, :392-578) between the batched searches, ON the
 * device, each reading the previous step's results and writing the next step's jobs -
 * a CU state is enqueued once and read back once (no host round trip inside):
 *
 *   xvcgpu_mc_metric_batch (SAD of both predictors)     -> d_start_dist
 *   xvcgpu_cs_start_fold   EvalStartMvp's choice        -> the search jobs' predictor
 *                          (+ previous_fullpel_, + the affine bootstrap vector)
 *   xvcgpu_me_search / xvcgpu_affine_me_batch           -> d_me_res / d_aff_res
 *   xvcgpu_cs_uni_fold     EvalFinalMvpIdx, SetMvd, GetInterPredBits (default bit
 *                          prices, include/xvc_inter_bits.h), the cost fold per list,
 *                          the list SearchBiIterative searches
 *                                                       -> the refinement job slots
 *   xvcgpu_bipred_search / xvcgpu_affine_me_batch       -> d_bi_res / d_aff_res
 *   xvcgpu_cs_bi_fold      the refinement's costs, the three-way choice (:247-257),
 *                          the affine pass against the plain one (:85-93),
 *                          HasZeroMvd                   -> d_results, the evaluation's
 *                                                          xvcgpu_inter_block jobs
 *
 * Refinement job slots of a pass: bi_job + (searched list * R + ref_idx) * R + other
 * ref_idx, R = XVC_CS_MAX_REFS; the fold fills the slots this state searches and empties
 * the others (block width 0: the search kernels return at once), so the host can issue
 * one launch per slot that exists without knowing which list won.  The affine pass's
 * slots and search jobs live in the xvcgpu_affine_me_block array.  A call folds the
 * passes [first, first + n) of d_passes / d_results (an affine pass names its plain
 * pass by absolute index; the plain pass must have been folded by an EARLIER launch of
 * the same fold - the affine pass reads and updates its result record -, i.e. a call
 * never holds a pass together with its plain pass).  A pass with
 * XVC_CS_FORCE_L1_MVD_ZERO or bi_iterations > 1 is answered XVC_CS_WHICH_UNSUPPORTED
 * (xvcgpu_types.h), not computed. */
xvcgpu_status xvcgpu_cs_start_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                   const uint64_t *d_start_dist, xvcgpu_me_block *d_me_jobs,
                                   const xvcgpu_me_result *d_me_res,
                                   xvcgpu_affine_me_block *d_aff_jobs,
                                   xvcgpu_cs_result *d_results, int pic_w, int pic_h);
xvcgpu_status xvcgpu_cs_uni_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                 const xvcgpu_me_result *d_me_res,
                                 const xvcgpu_affine_me_result *d_aff_res,
                                 xvcgpu_cs_result *d_results, xvcgpu_bi_block *d_bi_jobs,
                                 xvcgpu_affine_me_block *d_aff_jobs);
xvcgpu_status xvcgpu_cs_bi_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                const xvcgpu_me_result *d_bi_res,
                                const xvcgpu_affine_me_result *d_aff_res,
                                xvcgpu_cs_result *d_results, xvcgpu_inter_block *d_ev_inter);

/* The merge ranking's fold (xvcgpu_types.h: xvcgpu_cs_merge): rankings [first, first + n)
 * of d_merges / d_results; d_dist the SATDs, d_cands the ranking's prediction jobs,
 * d_ev_inter the evaluation slots.  One launch, no host decision. */
xvcgpu_status xvcgpu_cs_merge_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_merge *d_merges, int first,
                                   int n, const uint64_t *d_dist,
                                   const xvcgpu_inter_block *d_cands,
                                   xvcgpu_cs_merge_result *d_results,
                                   xvcgpu_inter_block *d_ev_inter);

/* ---- many chains, one launch per step kind ---------------------------------------- *
 * The pictures of one chain (the picture being coded, its reference pictures, the three
 * scratch pictures of its evaluations, its level and result arrays) as a device-resident
 * record; xvcgpu_cs_segs_launch then runs n_segs segments of one kind - the same kernel
 * bodies as the single-chain entry points named in xvcgpu_types.h, grid y = segment - on
 * ctx's stream.  Segments of one call must not depend on each other (they are the
 * current steps of DIFFERENT chains); a chain's next step goes into a later call.
 * All segments of a call share i0 where it selects a kernel instance (ME / BI / AFFINE).
 * The kernels read the segment records from page-locked memory of the context (a ring of
 * two 4 MiB halves; a half is written again only after the launches that read it have
 * passed), so a call takes any number of segments. */
xvcgpu_status xvcgpu_cs_env_create(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *const *refs, int n_refs,
                                   xvcgpu_picture *s_orig, xvcgpu_picture *s_pred,
                                   xvcgpu_picture *s_rec, int16_t *d_levels,
                                   xvcgpu_cs_result *d_results, xvcgpu_cs_env **out);
void xvcgpu_cs_env_destroy(xvcgpu_cs_env *env);
xvcgpu_status xvcgpu_cs_segs_launch(xvcgpu_ctx *ctx, int kind, const xvcgpu_cs_seg *segs,
                                    int n_segs);
/* 1 in *done when everything recorded into ev has completed, without waiting */
xvcgpu_status xvcgpu_event_query(xvcgpu_event *ev, int *done);

/* All distortions of an evaluation (CompressAndEvalCbf, inter_search.cc:261-365) in one
 * launch: candidate i compares its block of `orig` with the same block of `pred`
 * (versus = 0: the cbf-zero distortion) or of `rec` (versus = 1: an alternative's
 * reconstruction) in component comp with `metric`, times its weight -
 * out[i] = (uint64)(dist * weight) as SampleMetric::CompareSample returns it
 * (sample_metric.cc:221-222).  xvcgpu_metric_batch is one component, one picture pair
 * and one weight per launch. */
xvcgpu_status xvcgpu_eval_dist_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *pred, const xvcgpu_picture *rec,
                                     int structural_strength, const xvcgpu_eval_cand *d_cands,
                                     int n, uint64_t *d_out);

/* ---- multi-GPU staging --------------------------------------------------- *
 * n device-to-device copies (descriptors in device memory) in one launch: packs
 * the row slabs / CU metadata rows a rank exchanges with its neighbours into
 * one contiguous buffer per exchange and unpacks what arrived, so that an
 * exchange is a single collective instead of a dozen point-to-point
 * operations (each of which costs ~10 us of host time through
 * torch.distributed).  Segments must not overlap each other. */
xvcgpu_status xvcgpu_copy_segments(xvcgpu_ctx *ctx, const xvcgpu_copy_segment *d_segments,
                                   int n);

/* ---- a picture per call ------------------------------------------------- *
 * The per-picture sequence of the entry points above behind one call (host
 * cost of one call instead of six; nothing new is computed): the phases set in
 * a->phases run in the order ENCODE (InterSearch::SearchMotion +
 * CompressAndEvalCbf for the uni-pred CUs of a->d_me), DEBLOCK_V, DEBLOCK_H
 * (DeblockingFilter::DeblockPicture, by rows so that a row shard can exchange
 * its halo in between), PAD (YuvPicture::PadBorder), SSD (ComparePicture parts).
 * Asynchronous on the context's stream like its parts. */
xvcgpu_status xvcgpu_frame_pass(xvcgpu_ctx *ctx, const xvcgpu_frame_pass_args *a,
                                int phases);

/* The frame passes of n independent pictures (the pictures the reference's
 * picture-level worker threads code at the same time, thread_encoder.cc:99-159)
 * with every kernel launched ONCE for all of them: the n motion searches run
 * beside each other, then the n transforms, ... - kernels of one kind share the
 * chip far better than kernels of different pictures' different stages do.
 * Same results as n calls of xvcgpu_frame_pass(ctxs[i], args[i], phases).
 * Everything is enqueued on ctxs[0]'s stream; ctxs[i] lends picture i its
 * scratch (its own stream must be idle or the same stream).  Batched for
 * 2 <= n <= 4 whole pictures of one size whose CUs are 8x8 ... 16x16
 * (scratch_rec given), all phases, QuantFast or the packed RDOQ pipeline;
 * anything else falls back to the n single calls, each on its own context. */
xvcgpu_status xvcgpu_frame_pass_multi(xvcgpu_ctx *const *ctxs,
                                      const xvcgpu_frame_pass_args *const *args, int n,
                                      int phases);

/* ---- tables (host side, no GPU needed) ---------------------------------- *
 * The 8-bit-fraction transform matrices the kernels use (transform_data.cc:
 * 109-796), for table-equality tests. out: size*size int16 row-major. */
xvcgpu_status xvcgpu_get_transform_matrix(int tx_type, int size, int16_t *out);

#ifdef __cplusplus
}
#endif
#endif /* XVCGPU_H_ */
