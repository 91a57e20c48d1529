// xvc_picture_engine.h -- what one rank does for the entries of the picture timeline
// (xvc_picture_schedule.h): the device side of picture-level parallelism, in C++.
//
// The reference codes the independent pictures of a sub-GOP on worker threads
// (xvc_enc_lib/thread_encoder.cc:99-159); here a worker is a picture SLOT of a GPU -
// a context with its own stream - and a rank's walk over the timeline is
//   encode    the hot-path frame pass (xvcgpu_frame_pass) of the picture against its
//             nearest list-0 reference on the slot's stream, after the events of the
//             pictures it reads; the reconstruction lands in the ring entry of the
//             picture, an event marks it ready
//   transfer  xvcgpu_comm_send_picture / _recv_picture (RCCL over xGMI) on the
//             communicator's stream, ordered by the same events
// Ring of picture buffers per rank: entry = picture index modulo the ring size
// (window + 2 sub-GOPs + 1).  Before an entry is overwritten the writer waits for the
// picture that wrote its previous content and for everything that read it.
// Python (xvc_amd/picture_parallel.py) allocates and binds; nothing of the walk is
// decided there.
#ifndef XVC_AMD_HOST_XVC_PICTURE_ENGINE_H_
#define XVC_AMD_HOST_XVC_PICTURE_ENGINE_H_

#include <cstdint>

#include "xvc_picture_schedule.h"
#include "xvcgpu.h"

extern "C" {

typedef struct xvc_picture_engine_desc {
  const xvc_schedule *schedule;
  int32_t rank;
  int32_t n_slots;
  xvcgpu_ctx *const *ctxs;                     // one per picture slot
  xvcgpu_frame_pass_args *const *slot_args;    // per slot: the frame pass's buffers
                                               // (orig / ref / rec / ref_poc set per picture)
  xvcgpu_comm *comm;                           // NULL: one rank
  const xvcgpu_picture *const *orig_of_picture;   // per picture of the schedule
  int32_t ring;
  xvcgpu_picture *const *recs;                 // `ring` reconstructions
  // optional: called after the encode of a picture has been enqueued (tests read the
  // ring entry back before it is overwritten)
  int (*after_encode)(void *user, int picture_index);
  // optional, used when comm == NULL on more than one rank: a host transport for tests
  // that put several ranks on one GPU (RCCL refuses that) - the picture of ring entry
  // `entry` shipped / received by the caller, ordered on ctxs[0]'s stream
  int (*host_send)(void *user, int entry, int dst_rank);
  int (*host_recv)(void *user, int entry, int src_rank);
  void *user;
} xvc_picture_engine_desc;

typedef struct xvc_picture_engine xvc_picture_engine;
xvc_picture_engine *xvc_host_picture_engine_create(const xvc_picture_engine_desc *desc);
void xvc_host_picture_engine_destroy(xvc_picture_engine *e);
// timeline entries [first_op, end_op) (end_op < 0: to the end) as this rank
int xvc_host_picture_engine_run(xvc_picture_engine *e, int first_op, int end_op);
// test hook: the ring's ordering rule without a device (see the .cc)
int xvc_host_picture_ring_claims(int ring, int n, const int32_t *indices, const int32_t *reader_ids,
                                 int32_t *entries_out, int32_t *waited_out);
// the picture a ring entry holds (-1: none) / pictures encoded so far
int xvc_host_picture_engine_holds(const xvc_picture_engine *e, int entry);
int xvc_host_picture_engine_encoded(const xvc_picture_engine *e);
}

#endif  // XVC_AMD_HOST_XVC_PICTURE_ENGINE_H_
