// xvc_shard_filter.h -- the in-loop filter of a picture sharded by CTU rows over
// the GPUs of a node, for REAL CU trees (binary splits down to 4-tall CUs):
// SURVEY 8e scheme (A), the ordered hand-off.
//
// deblocking_filter.cc:56-152 filters a picture in two passes.  Vertical edges
// (pass 0) never cross rows: every rank filters its own rows.  Horizontal edges
// (pass 1) sit on the 4-sample grid (:59-62); the edge at row y reads rows
// y-4 .. y+3 and writes y-3 .. y+2, so two edges 4 rows apart in one 4-column band
// interact and must be applied in increasing y (:243-312) - a CHAIN.  Chains only
// form where CUs are 4 tall (common.h:99).  The edge ON a shard boundary y0 reads
// four rows of each side, and a chain may run on from it into the lower shard.
//
//   rank r owns rows [y0, y1).  From the CU map it derives the cut S = y0 + D: the
//   first multiple of 4 below y0 at which NO 4-column band has candidate edges at
//   both S - 4 and S (D = 4 when the CUs under the boundary are at least 8 tall;
//   xvc_shard_chain_rows below).  Edges further than 4 rows apart touch disjoint
//   rows, so the edges >= S and the edges < S commute.  (The longest run of
//   4-spaced edges starting AT y0 is not enough: a band whose run ends earlier can
//   start a new one that straddles that cut - round 3's rule, found by the
//   advisor, test_cut_clears_every_band.)  Then:
//     1. pass 0 on [y0, y1)
//     2. pass 1 on the edges [y0 + D, y1): nothing they read or write is touched
//        by the boundary edge's chain (rank 0: from row 0)
//     3. rows [y1 - 4, y1) - final as far as this rank's own edges go - to rank
//        r + 1;  rows [y0 - 4, y0) from rank r - 1
//     4. pass 1 on the strip of edges [y0, y0 + D): the boundary edge and its
//        chain, in order, on the rows just received
//     5. rows [y0 - 4, y0) (the edge modified three of them; chroma: one of two)
//        back to rank r - 1;  rows [y1 - 4, y1) back from rank r + 1
//   Exact for every CU tree (no assumption on CU heights); two small exchanges
//   per picture instead of scheme B's one; no rank waits for more than its upper
//   neighbour's step 2.  Requires y0 + D <= y1 - 4 (a shard taller than its chain);
//   every rank checks EVERY boundary (xvc_shard_filter_plan: all ranks hold the CU map and
//   reach the same verdict before the first transfer).
//
// ChainRows is host arithmetic on the CU map (also what the engine-agnostic
// Python mirror xvc_amd/sharded.ShardedTreeFilter and its CPU tests use);
// ShardedTreeFilter::Run issues the steps on xvcgpu_deblock_rows and the native
// RCCL row transfers of xvcgpu_comm_* (no torch on the data path).
#ifndef XVC_AMD_HOST_XVC_SHARD_FILTER_H_
#define XVC_AMD_HOST_XVC_SHARD_FILTER_H_

#include <cstdint>

#include "xvcgpu.h"

extern "C" {
// D for the boundary at luma row y0 (a multiple of 4, 0 < y0 < pic_h): cu_map as
// xvcgpu_deblock takes it (one int32 per 4x4 luma cell, -1 = no CU).
int xvc_shard_chain_rows(const int32_t *cu_map, int map_stride, int pic_w, int pic_h, int y0);

// D of every boundary (d_top[r] for rank r's upper boundary, d_top[0] = 0; d_top may
// be NULL): XVCGPU_OK, or XVCGPU_UNSUPPORTED when some shard is shorter than the
// chain entering it - the same answer on every rank.
int xvc_shard_filter_plan(const int32_t *cu_map, int map_stride, int world, const int32_t *rows,
                   int32_t *d_top);

// Steps 1-5 on callbacks: pass(which, ya, yb) filters the edges of direction `which`
// (0 vertical, 1 horizontal) in rows [ya, yb); exchange(send_down) is step 3
// (send_down = 1: the last four rows go down, the four rows above the shard arrive)
// or step 5 (0: they travel back).  The plan (every boundary, on every rank) and the
// order of the steps are this function's; the CPU tests run the oracle over gloo
// through it, xvc_host_shard_filter_run the HIP kernels over RCCL.
typedef struct xvc_shard_filter_callbacks {
  void *user;
  int (*pass)(void *user, int which, int ya, int yb);
  int (*exchange)(void *user, int send_down);
} xvc_shard_filter_callbacks;
int xvc_shard_filter_run(const int32_t *cu_map, int map_stride, int rank, int world,
                         const int32_t *rows, const xvc_shard_filter_callbacks *cb);

// Steps 1-5 for one picture on rank `rank` of `world` (comm may be NULL when world
// == 1): rows[r] .. rows[r + 1] are rank r's rows (multiples of 16), d_* the
// device copies of the CU records / map, cu_map the host copy (planning).
// Asynchronous on the context's and the communicator's streams.
int xvc_host_shard_filter_run(xvcgpu_ctx *ctx, xvcgpu_comm *comm, int rank, int world,
                              const int32_t *rows, xvcgpu_picture *rec,
                              const xvcgpu_cu_info *d_cus, int n_cus, const int32_t *d_cu_map,
                              const int32_t *cu_map, int map_stride, int pic_is_bipred,
                              int beta_offset, int tc_offset);
}

#endif  // XVC_AMD_HOST_XVC_SHARD_FILTER_H_
