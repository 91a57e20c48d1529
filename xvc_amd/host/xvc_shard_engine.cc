// xvc_shard_engine.cc -- see xvc_shard_engine.h.
#include "xvc_shard_engine.h"

#include <algorithm>
#include <new>
#include <utility>
#include <vector>

struct xvc_shard_plan {
  int width, height, cu, world, rank, cus_per_row;
  int reach_up, reach_down;
  std::vector<std::pair<int, int>> rows;   // per rank [y0, y1)
  std::vector<xvc_shard_slab> slabs[2][2];  // [halo / gather][send / recv]
};

namespace {

const int kHalo = 4;   // luma rows on each side of a shard boundary

// the CU rows of a picture split into `world` contiguous shards, the first
// n_rows % world one row taller
void ShardRows(int height, int world, int cu, std::vector<std::pair<int, int>> *out) {
  const int n_rows = (height + cu - 1) / cu;
  const int base = n_rows / world, extra = n_rows % world;
  int r = 0;
  for (int k = 0; k < world; k++) {
    const int n = base + (k < extra ? 1 : 0);
    out->push_back(std::make_pair(r * cu, std::min(height, (r + n) * cu)));
    r += n;
  }
}

// rows of `peer`'s shard that rank `who` must hold for its next search / PSNR walk
bool NeededFrom(const xvc_shard_plan &p, int who, int peer, int *ya, int *yb) {
  *ya = std::max(p.rows[peer].first, p.rows[who].first - p.reach_up);
  *yb = std::min(p.rows[peer].second, p.rows[who].second + p.reach_down);
  return *ya < *yb;
}

void Add(std::vector<xvc_shard_slab> *v, int peer, int kind, int a, int b) {
  xvc_shard_slab s = {peer, kind, a, b};
  v->push_back(s);
}

}  // namespace

extern "C" {

int xvc_shard_rows(int height, int world, int cu, int r, int32_t *y0, int32_t *y1) {
  if (height <= 0 || cu <= 0 || world < 1 || r < 0 || r >= world ||
      world > (height + cu - 1) / cu)
    return XVCGPU_INVALID_ARGUMENT;
  std::vector<std::pair<int, int>> rows;
  ShardRows(height, world, cu, &rows);
  *y0 = rows[r].first;
  *y1 = rows[r].second;
  return XVCGPU_OK;
}

xvc_shard_plan *xvc_shard_plan_create(int width, int height, int cu, int world, int rank,
                                      int reach, int min_cu_h_top, int min_cu_h_bottom) {
  if (width <= 0 || height <= 0 || cu < 8 || world < 1 || rank < 0 || rank >= world ||
      world > (height + cu - 1) / cu || reach < 0)
    return nullptr;
  xvc_shard_plan *p = new (std::nothrow) xvc_shard_plan();
  if (!p) return nullptr;
  p->width = width;
  p->height = height;
  p->cu = cu;
  p->world = world;
  p->rank = rank;
  p->cus_per_row = (width + cu - 1) / cu;
  ShardRows(height, world, cu, &p->rows);
  p->reach_up = (reach + 15) / 16 * 16;
  p->reach_down = std::max(p->reach_up, 64);   // + the 64-row PSNR blocks that start in the own rows
  const int y0 = p->rows[rank].first, y1 = p->rows[rank].second;
  const bool up = rank > 0, down = rank < world - 1;
  // a 4-tall CU at a boundary starts a chain of interacting edges across it: the
  // redundant edge of scheme B would be filtered out of order
  if ((up && min_cu_h_top < 8) || (down && min_cu_h_bottom < 8)) {
    delete p;
    return nullptr;
  }
  const int per_row = p->cus_per_row;
  std::vector<xvc_shard_slab> *hs = &p->slabs[0][0], *hr = &p->slabs[0][1];
  if (up) {
    Add(hs, rank - 1, XVC_SLAB_ROWS, y0, y0 + kHalo);
    Add(hs, rank - 1, XVC_SLAB_CUS, (y0 / cu) * per_row, per_row);
    Add(hr, rank - 1, XVC_SLAB_ROWS, y0 - kHalo, y0);
    Add(hr, rank - 1, XVC_SLAB_CUS, (y0 / cu - 1) * per_row, per_row);
  }
  if (down) {
    Add(hs, rank + 1, XVC_SLAB_ROWS, y1 - kHalo, y1);
    Add(hs, rank + 1, XVC_SLAB_CUS, (y1 / cu - 1) * per_row, per_row);
    Add(hr, rank + 1, XVC_SLAB_ROWS, y1, y1 + kHalo);
    Add(hr, rank + 1, XVC_SLAB_CUS, (y1 / cu) * per_row, per_row);
  }
  for (int peer = 0; peer < world; peer++) {
    if (peer == rank) continue;
    int ya, yb;
    if (NeededFrom(*p, peer, rank, &ya, &yb)) Add(&p->slabs[1][0], peer, XVC_SLAB_ROWS, ya, yb);
    if (NeededFrom(*p, rank, peer, &ya, &yb)) Add(&p->slabs[1][1], peer, XVC_SLAB_ROWS, ya, yb);
  }
  return p;
}

void xvc_shard_plan_destroy(xvc_shard_plan *plan) { delete plan; }

void xvc_shard_plan_rows(const xvc_shard_plan *plan, int r, int32_t *y0, int32_t *y1) {
  *y0 = plan->rows[r].first;
  *y1 = plan->rows[r].second;
}

void xvc_shard_plan_valid_rows(const xvc_shard_plan *plan, int32_t *ya, int32_t *yb) {
  *ya = std::max(0, plan->rows[plan->rank].first - plan->reach_up);
  *yb = std::min(plan->height, plan->rows[plan->rank].second + plan->reach_down);
}

int xvc_shard_plan_slabs(const xvc_shard_plan *plan, int which, int dir,
                         const xvc_shard_slab **out) {
  if (!plan || which < 0 || which > 1 || dir < 0 || dir > 1) return -1;
  const std::vector<xvc_shard_slab> &v = plan->slabs[which][dir];
  if (out) *out = v.empty() ? nullptr : &v[0];
  return static_cast<int>(v.size());
}

void xvc_shard_plan_traffic(const xvc_shard_plan *plan, int which, int64_t *messages,
                            int64_t *bytes) {
  // the library's picture layout: borders 128 / 64, strides rounded up to 64 samples
  const int64_t ls = (plan->width + 2 * XVCGPU_BORDER_LUMA + 63) / 64 * 64;
  const int64_t cs = (plan->width / 2 + 2 * XVCGPU_BORDER_CHROMA + 63) / 64 * 64;
  int64_t m = 0, b = 0;
  const std::vector<xvc_shard_slab> &v = plan->slabs[which][0];
  for (size_t i = 0; i < v.size(); i++) {
    if (v[i].kind == XVC_SLAB_ROWS) {
      m += 3;
      b += 2 * (ls * (v[i].b - v[i].a) + 2 * cs * ((v[i].b - v[i].a) / 2));
    } else {
      m += 1;
      b += static_cast<int64_t>(sizeof(xvcgpu_cu_info)) * v[i].b;
    }
  }
  *messages = m;
  *bytes = b;
}

int xvc_shard_run(const xvc_shard_plan *plan, const xvc_shard_callbacks *cb) {
  if (!plan || !cb || !cb->phase || !cb->exchange) return XVCGPU_INVALID_ARGUMENT;
  const int y0 = plan->rows[plan->rank].first, y1 = plan->rows[plan->rank].second;
  const bool down = plan->rank < plan->world - 1;
  int st = cb->phase(cb->user, 0, y0, y1, y1);
  if (st) return st;
  for (int which = 0; which < 2; which++) {
    const std::vector<xvc_shard_slab> &s = plan->slabs[which][0], &r = plan->slabs[which][1];
    // with more than one rank every shard has a neighbour; a single rank has nothing to swap
    if (plan->world > 1 || !s.empty() || !r.empty()) {
      st = cb->exchange(cb->user, which, s.empty() ? nullptr : &s[0], static_cast<int>(s.size()),
                        r.empty() ? nullptr : &r[0], static_cast<int>(r.size()));
      if (st) return st;
    }
    if (which == 0) {
      st = cb->phase(cb->user, 1, y0, y1, down ? y1 + kHalo : y1);
      if (st) return st;
    }
  }
  return cb->phase(cb->user, 2, y0, y1, y1);
}

// ---- the product: xvcgpu_frame_pass on row ranges + RCCL ---------------------------
namespace {

int GpuPhase(void *user, int which, int y0, int y1, int y_end) {
  xvc_shard_gpu *g = static_cast<xvc_shard_gpu *>(user);
  xvcgpu_frame_pass_args a = *g->args;
  int phases;
  if (which == 0) {
    a.db_y_begin = y0;
    a.db_y_end = y1;
    a.dbh_y_end = y1;
    phases = XVC_FP_ENCODE | XVC_FP_DEBLOCK_V;
  } else if (which == 1) {
    a.db_y_begin = y0;
    a.db_y_end = y_end;
    a.dbh_y_end = y_end;
    phases = XVC_FP_DEBLOCK_H;
  } else {
    a.ssd_y_begin = y0;
    a.ssd_y_end = y1;
    phases = XVC_FP_PAD | XVC_FP_SSD;
  }
  return xvcgpu_frame_pass(g->ctx, &a, phases);
}

int GpuExchange(void *user, int, const xvc_shard_slab *sends, int ns, const xvc_shard_slab *recvs,
                int nr) {
  xvc_shard_gpu *g = static_cast<xvc_shard_gpu *>(user);
  if (!ns && !nr) return XVCGPU_OK;
  if (!g->comm) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_status st;
  // the slabs are final / free to overwrite once the kernels queued so far are done
  if ((st = xvcgpu_event_record(g->ctx, g->before)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_wait_event(g->comm, g->before)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_group_begin(g->comm)) != XVCGPU_OK) return st;
  st = XVCGPU_OK;
  for (int i = 0; i < ns && st == XVCGPU_OK; i++)
    st = sends[i].kind == XVC_SLAB_ROWS
             ? xvcgpu_comm_send_rows(g->comm, g->args->rec, 7, sends[i].a, sends[i].b, sends[i].peer)
             : xvcgpu_comm_send_bytes(g->comm, g->d_cus + sends[i].a,
                                      sizeof(xvcgpu_cu_info) * sends[i].b, sends[i].peer);
  for (int i = 0; i < nr && st == XVCGPU_OK; i++)
    st = recvs[i].kind == XVC_SLAB_ROWS
             ? xvcgpu_comm_recv_rows(g->comm, g->args->rec, 7, recvs[i].a, recvs[i].b, recvs[i].peer)
             : xvcgpu_comm_recv_bytes(g->comm, g->d_cus + recvs[i].a,
                                      sizeof(xvcgpu_cu_info) * recvs[i].b, recvs[i].peer);
  const xvcgpu_status end = xvcgpu_comm_group_end(g->comm);   // closed on every path
  if (st != XVCGPU_OK) return st;
  if (end != XVCGPU_OK) return end;
  if ((st = xvcgpu_comm_record_event(g->comm, g->after)) != XVCGPU_OK) return st;
  return xvcgpu_event_wait(g->ctx, g->after);   // later kernels see the received rows
}

}  // namespace

int xvc_host_sharded_frame_pass(const xvc_shard_plan *plan, xvc_shard_gpu *gpu) {
  if (!plan || !gpu || !gpu->ctx || !gpu->args || !gpu->d_cus || !gpu->before || !gpu->after ||
      (plan->world > 1 && !gpu->comm))
    return XVCGPU_INVALID_ARGUMENT;
  if (gpu->comm && (xvcgpu_comm_world(gpu->comm) != plan->world ||
                    xvcgpu_comm_rank(gpu->comm) != plan->rank))
    return XVCGPU_INVALID_ARGUMENT;
  xvc_shard_callbacks cb = {gpu, GpuPhase, GpuExchange};
  return xvc_shard_run(plan, &cb);
}

int xvc_host_sharded_total_ssd(const xvc_shard_plan *plan, xvc_shard_gpu *gpu, uint64_t *d_ssd) {
  if (!plan || !gpu || !gpu->ctx || !d_ssd) return XVCGPU_INVALID_ARGUMENT;
  if (plan->world == 1 || !gpu->comm) return xvcgpu_sync(gpu->ctx);
  xvcgpu_status st;
  if ((st = xvcgpu_event_record(gpu->ctx, gpu->before)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_wait_event(gpu->comm, gpu->before)) != XVCGPU_OK) return st;
  if ((st = xvcgpu_comm_all_reduce_sum_u64(gpu->comm, d_ssd, 2)) != XVCGPU_OK) return st;
  return xvcgpu_comm_sync(gpu->comm);
}

}  // extern "C"
