// xvc_shard_engine.h -- the frame pass of a picture sharded by rows of CUs over the
// GPUs of one node: plan and control in C++ (north star: "host code stays C++ and
// calls HIP through a thin C-ABI layer ... frames shard by CTU rows across the 8
// GPUs of one node with RCCL halo exchange over xGMI for in-loop filtering").
//
// Per picture, on every rank (scheme B; exact while no 4-tall CU touches
// a shard boundary - pictures with real CU trees take xvc_shard_filter.h's ordered
// hand-off for their filter):
//   A   search, CompressAndEvalCbf and CU records of the own CU rows; deblocking of
//       the vertical edges (deblocking_filter.cc:56-152 pass 1 never crosses rows)
//   X1  halo: the HALO = 4 luma (2 chroma) rows either side of each boundary and the
//       CU records of the boundary CU rows swap with the neighbour ranks
//   B   horizontal edges of the own rows and of the first edge row of the shard below
//       (both neighbours compute that edge: same inputs, same result, no return)
//   X2  gather: finished rows go to the ranks whose next search can reach them - a
//       shard's search reads the reference within `reach` rows of its own - so on a
//       tall picture only neighbours exchange rows (not an all-gather)
//   C   PadBorder; the PSNR walk over the 64-row blocks that start in the own rows
//
// Three layers:
//   xvc_shard_plan     which rows / records go where (pure arithmetic; every rank
//                      derives the same plan)
//   xvc_shard_run      the five steps on callbacks - the engine (phases) and the
//                      transport (exchanges) are the caller's: the CPU tests run the
//                      oracle over gloo through this very control
//   xvc_host_sharded_frame_pass   the product: engine = xvcgpu_frame_pass on row
//                      ranges, transport = ncclSend / ncclRecv groups on the
//                      communicator's stream (xvcgpu_comm_*), ordered with the
//                      kernels by two events; no torch on the data path
#ifndef XVC_AMD_HOST_XVC_SHARD_ENGINE_H_
#define XVC_AMD_HOST_XVC_SHARD_ENGINE_H_

#include <cstdint>

#include "xvcgpu.h"

extern "C" {

enum { XVC_SLAB_ROWS = 0, XVC_SLAB_CUS = 1 };
// One piece of an exchange.  ROWS: luma rows [a, b) of all three planes (chroma
// rows a/2 .. b/2), full padded width; CUS: b CU records from record a.
typedef struct xvc_shard_slab {
  int32_t peer, kind, a, b;
} xvc_shard_slab;

typedef struct xvc_shard_plan xvc_shard_plan;

// rows [y0, y1) of rank r when the CU rows of a picture are split into `world`
// contiguous shards (the first n_rows % world one CU row taller)
int xvc_shard_rows(int height, int world, int cu, int r, int32_t *y0, int32_t *y1);

// reach: rows of the reference a shard's search may touch beyond its own rows
// (search range + predictor offset + clip margin + filter taps; the same number on
// every rank).  min_cu_h_top / _bottom: the smallest CU height at the shard's upper /
// lower boundary (64 where there is none): < 8 refuses the plan (NULL) - scheme B's
// exactness precondition.
xvc_shard_plan *xvc_shard_plan_create(int width, int height, int cu, int world, int rank,
                                      int reach, int min_cu_h_top, int min_cu_h_bottom);
void xvc_shard_plan_destroy(xvc_shard_plan *plan);
// rows of rank r: y0, y1 (luma lines)
void xvc_shard_plan_rows(const xvc_shard_plan *plan, int r, int32_t *y0, int32_t *y1);
// rows of the local reconstruction that are up to date after a picture
void xvc_shard_plan_valid_rows(const xvc_shard_plan *plan, int32_t *ya, int32_t *yb);
// the slabs of exchange `which` (0 halo, 1 gather) in direction `dir` (0 send, 1
// receive); per peer the order of one side's sends is the order of the other's receives
int xvc_shard_plan_slabs(const xvc_shard_plan *plan, int which, int dir,
                         const xvc_shard_slab **out);
// what one picture moves: messages (RCCL operations: a ROWS slab is three) and bytes
// this rank sends in exchange `which`, for a picture of the library's layout
void xvc_shard_plan_traffic(const xvc_shard_plan *plan, int which, int64_t *messages,
                            int64_t *bytes);

typedef struct xvc_shard_callbacks {
  void *user;
  // which: 0 = A on rows [y0, y1); 1 = B: horizontal edges of [y0, y_end); 2 = C:
  // border + PSNR parts of the blocks starting in [y0, y1)
  int (*phase)(void *user, int which, int y0, int y1, int y_end);
  // which: 0 halo, 1 gather
  int (*exchange)(void *user, int which, const xvc_shard_slab *sends, int n_sends,
                  const xvc_shard_slab *recvs, int n_recvs);
} xvc_shard_callbacks;
// A, X1, B, X2, C.  Returns the first non-zero callback result.
int xvc_shard_run(const xvc_shard_plan *plan, const xvc_shard_callbacks *cb);

// The product path.  `args`: the picture's frame-pass arguments as for an unsharded
// xvcgpu_frame_pass of the own CUs (orig, ref, rec, the own CUs' jobs, the whole
// picture's CU records / map, d_ssd ...); the row ranges are set here per phase.
// comm may be NULL when world == 1.  Asynchronous on the context's and the
// communicator's streams; before / after: two events of the context.
typedef struct xvc_shard_gpu {
  xvcgpu_ctx *ctx;
  xvcgpu_comm *comm;
  xvcgpu_frame_pass_args *args;
  xvcgpu_cu_info *d_cus;        // the whole picture's CU records (boundary rows travel)
  xvcgpu_event *before, *after;
} xvc_shard_gpu;
int xvc_host_sharded_frame_pass(const xvc_shard_plan *plan, xvc_shard_gpu *gpu);
// sum of the ranks' PSNR parts (two uint64 at d_ssd), in place; synchronises
int xvc_host_sharded_total_ssd(const xvc_shard_plan *plan, xvc_shard_gpu *gpu, uint64_t *d_ssd);
}

#endif  // XVC_AMD_HOST_XVC_SHARD_ENGINE_H_
