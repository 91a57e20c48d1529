// xvc_shard_filter.cc -- see xvc_shard_filter.h.
#include "xvc_shard_filter.h"

namespace {

struct Events {
  xvcgpu_ctx *ctx;
  xvcgpu_event *a, *b;
  explicit Events(xvcgpu_ctx *c) : ctx(c), a(nullptr), b(nullptr) {}
  ~Events() {
    if (a) xvcgpu_event_destroy(a);
    if (b) xvcgpu_event_destroy(b);
  }
};

}  // namespace

extern "C" {

int xvc_shard_chain_rows(const int32_t *cu_map, int map_stride, int pic_w, int pic_h, int y0) {
  if (!cu_map || y0 <= 0 || y0 >= pic_h || (y0 & 3)) return 0;
  const int cols = (pic_w + 3) / 4, rows = (pic_h + 3) / 4;
  // candidate edge at cell row cy: the cell and the cell above belong to different
  // CUs (deblocking_filter.cc:98-114)
  auto edge = [&](int cy, int cx) {
    const int32_t q = cu_map[cy * map_stride + cx], p = cu_map[(cy - 1) * map_stride + cx];
    return q >= 0 && p >= 0 && p != q;
  };
  // The cut S = y0 + D separates the edges applied early (step 2: >= S) from the
  // strip applied late (step 4: [y0, S)).  Only edges 4 rows apart interact (an
  // edge writes y-3 .. y+2 and reads y-4 .. y+3), so the order of the two sets is
  // immaterial exactly when NO band has candidate edges at both S - 4 and S: the
  // first such S below the boundary (a band whose own run ends earlier than the
  // longest one may start a new run before the cut of the longest - one cut for
  // the whole boundary has to clear every band).
  for (int cs = y0 / 4 + 1; cs < rows; cs++) {
    bool clash = false;
    for (int cx = 0; cx < cols && !clash; cx++) clash = edge(cs - 1, cx) && edge(cs, cx);
    if (!clash) return 4 * cs - y0;
  }
  return 4 * rows - y0;
}

int xvc_shard_filter_plan(const int32_t *cu_map, int map_stride, int world, const int32_t *rows,
                   int32_t *d_top) {
  if (!cu_map || !rows || world < 1) return XVCGPU_INVALID_ARGUMENT;
  const int pic_h = rows[world], pic_w = map_stride * 4;
  int verdict = XVCGPU_OK;
  for (int r = 0; r < world; r++) {
    if (rows[r + 1] <= rows[r] || (rows[r] & 3)) return XVCGPU_INVALID_ARGUMENT;
    const int d = r > 0 ? xvc_shard_chain_rows(cu_map, map_stride, pic_w, pic_h, rows[r]) : 0;
    if (d_top) d_top[r] = d;
    // a shard has to be taller than the chain that enters it (the last one only
    // has to hold its boundary edge's rows)
    if (r < world - 1 ? rows[r] + d > rows[r + 1] - 4 : (r > 0 && rows[r + 1] - rows[r] < 4))
      verdict = XVCGPU_UNSUPPORTED;
  }
  return verdict;
}

int xvc_shard_filter_run(const int32_t *cu_map, int map_stride, int rank, int world,
                         const int32_t *rows, const xvc_shard_filter_callbacks *cb) {
  if (!cu_map || !rows || !cb || !cb->pass || !cb->exchange || world < 1 || rank < 0 ||
      rank >= world || world > 64)
    return XVCGPU_INVALID_ARGUMENT;
  // Every rank holds the whole CU map: all of them plan ALL boundaries and reach
  // the same verdict before anything is sent (a rank that backed out alone would
  // leave its neighbours waiting in their send / receive groups).
  int32_t d_all[64];
  {
    const int st = xvc_shard_filter_plan(cu_map, map_stride, world, rows, d_all);
    if (st != XVCGPU_OK) return st;
  }
  const int y0 = rows[rank], y1 = rows[rank + 1], d_top = d_all[rank];
  int st;
  if ((st = cb->pass(cb->user, 0, y0, y1)) != 0) return st;                    // 1
  if (y0 + d_top < y1 && (st = cb->pass(cb->user, 1, y0 + d_top, y1)) != 0) return st;   // 2
  if (world == 1) return XVCGPU_OK;
  if ((st = cb->exchange(cb->user, 1)) != 0) return st;                        // 3
  // 4: the boundary edge and its chain, in order, on the rows just received
  if (rank > 0 && d_top > 0 && (st = cb->pass(cb->user, 1, y0, y0 + d_top)) != 0) return st;
  return cb->exchange(cb->user, 0);                                            // 5
}

namespace {
struct GpuFilter {
  xvcgpu_ctx *ctx;
  xvcgpu_comm *comm;
  int rank, world, y0, y1;
  xvcgpu_picture *rec;
  const xvcgpu_cu_info *d_cus;
  int n_cus;
  const int32_t *d_cu_map;
  int map_stride, bipred, beta, tc;
  xvcgpu_event *a, *b;
};
int GpuFilterPass(void *user, int which, int ya, int yb) {
  GpuFilter *g = static_cast<GpuFilter *>(user);
  if (ya >= yb) return XVCGPU_OK;
  return xvcgpu_deblock_rows(g->ctx, g->rec, g->d_cus, g->n_cus, g->d_cu_map, g->map_stride,
                             g->bipred, g->beta, g->tc, 4, which, ya, yb);
}
// an error between group_begin and group_end must not leave the RCCL group open
int GpuFilterExchange(void *user, int send_down) {
  GpuFilter *g = static_cast<GpuFilter *>(user);
  const bool up = g->rank > 0, down = g->rank < g->world - 1;
  xvcgpu_status st;
  if ((st = xvcgpu_event_record(g->ctx, g->a)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_wait_event(g->comm, g->a)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_group_begin(g->comm)) != XVCGPU_OK) return st;
  st = XVCGPU_OK;
  if (send_down) {  // 3: rows down / rows from above
    if (down) st = xvcgpu_comm_send_rows(g->comm, g->rec, 7, g->y1 - 4, g->y1, g->rank + 1);
    if (st == XVCGPU_OK && up)
      st = xvcgpu_comm_recv_rows(g->comm, g->rec, 7, g->y0 - 4, g->y0, g->rank - 1);
  } else {  // 5: the rows the boundary edge changed go back up / come back from below
    if (up) st = xvcgpu_comm_send_rows(g->comm, g->rec, 7, g->y0 - 4, g->y0, g->rank - 1);
    if (st == XVCGPU_OK && down)
      st = xvcgpu_comm_recv_rows(g->comm, g->rec, 7, g->y1 - 4, g->y1, g->rank + 1);
  }
  const xvcgpu_status end = xvcgpu_comm_group_end(g->comm);
  if (st != XVCGPU_OK) return st;
  if (end != XVCGPU_OK) return end;
  if ((st = xvcgpu_comm_record_event(g->comm, g->b)) != XVCGPU_OK) return st;
  return xvcgpu_event_wait(g->ctx, g->b);
}
}  // namespace

int xvc_host_shard_filter_run(xvcgpu_ctx *ctx, xvcgpu_comm *comm, int rank, int world,
                              const int32_t *rows, xvcgpu_picture *rec,
                              const xvcgpu_cu_info *d_cus, int n_cus, const int32_t *d_cu_map,
                              const int32_t *cu_map, int map_stride, int pic_is_bipred,
                              int beta_offset, int tc_offset) {
  if (!ctx || !rows || !rec || !d_cus || !d_cu_map || !cu_map || world < 1 || rank < 0 ||
      rank >= world || (world > 1 && !comm))
    return XVCGPU_INVALID_ARGUMENT;
  Events ev(ctx);
  if (world > 1) {
    xvcgpu_status st;
    if ((st = xvcgpu_event_create(ctx, &ev.a)) != XVCGPU_OK) return st;
    if ((st = xvcgpu_event_create(ctx, &ev.b)) != XVCGPU_OK) return st;
  }
  GpuFilter g = {ctx, comm, rank, world, rows[rank], rows[rank + 1], rec, d_cus, n_cus, d_cu_map,
                 map_stride, pic_is_bipred, beta_offset, tc_offset, ev.a, ev.b};
  const xvc_shard_filter_callbacks cb = {&g, GpuFilterPass, GpuFilterExchange};
  const int st = xvc_shard_filter_run(cu_map, map_stride, rank, world, rows, &cb);
  if (st != XVCGPU_OK || world == 1) return st;
  // the events may be destroyed once enqueued work has passed them; keep the
  // host in step with the two short exchanges (a picture's worth of filtering)
  return xvcgpu_comm_sync(comm);
}

}  // extern "C"
