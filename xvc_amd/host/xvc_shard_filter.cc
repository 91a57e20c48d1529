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
  int longest = 0;
  for (int cx = 0; cx < cols; cx++) {
    int run = 0;
    // candidate edge at row y: the cell and the cell above belong to different
    // CUs (deblocking_filter.cc:98-114)
    for (int cy = y0 / 4; cy < rows; cy++) {
      const int32_t q = cu_map[cy * map_stride + cx], p = cu_map[(cy - 1) * map_stride + cx];
      if (q < 0 || p < 0 || p == q) break;
      run++;
    }
    if (run > longest) longest = run;
  }
  return 4 * (longest + 1);
}

int xvc_host_shard_filter_run(xvcgpu_ctx *ctx, xvcgpu_comm *comm, int rank, int world,
                              const int32_t *rows, xvcgpu_picture *rec,
                              const xvcgpu_cu_info *d_cus, int n_cus, const int32_t *d_cu_map,
                              const int32_t *cu_map, int map_stride, int pic_is_bipred,
                              int beta_offset, int tc_offset) {
  if (!ctx || !rows || !rec || !d_cus || !d_cu_map || !cu_map || world < 1 || rank < 0 ||
      rank >= world || (world > 1 && !comm))
    return XVCGPU_INVALID_ARGUMENT;
  const int pic_h = rows[world];
  const int y0 = rows[rank], y1 = rows[rank + 1];
  int pic_w = 0, ph = 0, bd = 0;
  // (the planning only needs the width in cells: the map's stride bounds it)
  pic_w = map_stride * 4;
  (void)ph;
  (void)bd;
  const int d_top = rank > 0 ? xvc_shard_chain_rows(cu_map, map_stride, pic_w, pic_h, y0) : 0;
  if (y0 + d_top > y1 - 4 && rank < world - 1) return XVCGPU_UNSUPPORTED;  // shard shorter than its chain
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
  // 3: rows down / rows from above, one RCCL group on the communicator's stream
  TRY(xvcgpu_event_record(ctx, ev.a));
  TRY(xvcgpu_comm_wait_event(comm, ev.a));
  TRY(xvcgpu_comm_group_begin(comm));
  if (down) TRY(xvcgpu_comm_send_rows(comm, rec, 7, y1 - 4, y1, rank + 1));
  if (up) TRY(xvcgpu_comm_recv_rows(comm, rec, 7, y0 - 4, y0, rank - 1));
  TRY(xvcgpu_comm_group_end(comm));
  TRY(xvcgpu_comm_record_event(comm, ev.b));
  TRY(xvcgpu_event_wait(ctx, ev.b));
  // 4: the boundary edge and its chain
  if (up) TRY(pass(1, y0, y0 + d_top));
  // 5: the rows the boundary edge changed go back up / come back from below
  TRY(xvcgpu_event_record(ctx, ev.a));
  TRY(xvcgpu_comm_wait_event(comm, ev.a));
  TRY(xvcgpu_comm_group_begin(comm));
  if (up) TRY(xvcgpu_comm_send_rows(comm, rec, 7, y0 - 4, y0, rank - 1));
  if (down) TRY(xvcgpu_comm_recv_rows(comm, rec, 7, y1 - 4, y1, rank + 1));
  TRY(xvcgpu_comm_group_end(comm));
  TRY(xvcgpu_comm_record_event(comm, ev.b));
  TRY(xvcgpu_event_wait(ctx, ev.b));
  // the events may be destroyed once enqueued work has passed them; keep the
  // host in step with the two short exchanges (a picture's worth of filtering)
  TRY(xvcgpu_comm_sync(comm));
#undef TRY
  return XVCGPU_OK;
}

}  // extern "C"
