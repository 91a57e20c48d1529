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

int xvc_shard_plan(const int32_t *cu_map, int map_stride, int world, const int32_t *rows,
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

int xvc_host_shard_filter_run(xvcgpu_ctx *ctx, xvcgpu_comm *comm, int rank, int world,
                              const int32_t *rows, xvcgpu_picture *rec,
                              const xvcgpu_cu_info *d_cus, int n_cus, const int32_t *d_cu_map,
                              const int32_t *cu_map, int map_stride, int pic_is_bipred,
                              int beta_offset, int tc_offset) {
  if (!ctx || !rows || !rec || !d_cus || !d_cu_map || !cu_map || world < 1 || rank < 0 ||
      rank >= world || (world > 1 && !comm))
    return XVCGPU_INVALID_ARGUMENT;
  const int y0 = rows[rank], y1 = rows[rank + 1];
  // Every rank holds the whole CU map: all of them plan ALL boundaries and reach
  // the same verdict before anything is sent (a rank that backed out alone would
  // leave its neighbours waiting in their send / receive groups).
  int32_t d_all[64];
  if (world > 64) return XVCGPU_INVALID_ARGUMENT;
  {
    const int st = xvc_shard_plan(cu_map, map_stride, world, rows, d_all);
    if (st != XVCGPU_OK) return st;
  }
  const int d_top = d_all[rank];
#define TRY(call)                                  \
  do {                                             \
    const xvcgpu_status st_ = (call);              \
    if (st_ != XVCGPU_OK) return st_;              \
  } while (0)
  auto pass = [&](int which, int ya, int yb) -> xvcgpu_status {
    if (ya >= yb) return XVCGPU_OK;
    return xvcgpu_deblock_rows(ctx, rec, d_cus, n_cus, d_cu_map, map_stride, pic_is_bipred,
                               beta_offset, tc_offset, 4, which, ya, yb);
  };
  TRY(pass(0, y0, y1));                    // 1
  TRY(pass(1, y0 + d_top, y1));            // 2
  if (world == 1) return XVCGPU_OK;
  Events ev(ctx);
  TRY(xvcgpu_event_create(ctx, &ev.a));
  TRY(xvcgpu_event_create(ctx, &ev.b));
  const bool up = rank > 0, down = rank < world - 1;
  // an error between group_begin and group_end must not leave the RCCL group open
  auto exchange = [&](bool send_down) -> xvcgpu_status {
    TRY(xvcgpu_event_record(ctx, ev.a));
    TRY(xvcgpu_comm_wait_event(comm, ev.a));
    TRY(xvcgpu_comm_group_begin(comm));
    xvcgpu_status st = XVCGPU_OK;
    if (send_down) {  // 3: rows down / rows from above
      if (down) st = xvcgpu_comm_send_rows(comm, rec, 7, y1 - 4, y1, rank + 1);
      if (st == XVCGPU_OK && up) st = xvcgpu_comm_recv_rows(comm, rec, 7, y0 - 4, y0, rank - 1);
    } else {  // 5: the rows the boundary edge changed go back up / come back from below
      if (up) st = xvcgpu_comm_send_rows(comm, rec, 7, y0 - 4, y0, rank - 1);
      if (st == XVCGPU_OK && down) st = xvcgpu_comm_recv_rows(comm, rec, 7, y1 - 4, y1, rank + 1);
    }
    const xvcgpu_status end = xvcgpu_comm_group_end(comm);
    if (st != XVCGPU_OK) return st;
    TRY(end);
    TRY(xvcgpu_comm_record_event(comm, ev.b));
    return xvcgpu_event_wait(ctx, ev.b);
  };
  TRY(exchange(true));
  // 4: the boundary edge and its chain
  if (up) TRY(pass(1, y0, y0 + d_top));
  TRY(exchange(false));
  // the events may be destroyed once enqueued work has passed them; keep the
  // host in step with the two short exchanges (a picture's worth of filtering)
  TRY(xvcgpu_comm_sync(comm));
#undef TRY
  return XVCGPU_OK;
}

}  // extern "C"
