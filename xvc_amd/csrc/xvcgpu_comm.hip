// xvcgpu_comm.hip -- the multi-GPU exchange of the hot path, issued natively on
// RCCL over xGMI: one communicator per process (= per GPU), point-to-point
// transfers of padded reference pictures between the ranks that code
// independent pictures (picture-level sharding, the way
// ThreadEncoder scales, thread_encoder.cc:99-159) and of plane rows between
// CTU-row shards (deblocking halo, deblocking_filter.cc:59-62).
//
// Streams: a communicator has its own HIP stream, so transfers run beside the
// kernels of the contexts.  Ordering is by events (xvcgpu_event_*): a send
// waits for the event recorded after the picture was coded, a receive records
// the event its consumers wait for.  ncclSend / ncclRecv pairs that are
// enqueued in the same order on every rank cannot deadlock; pairs that may
// cross go between xvcgpu_comm_group_begin / _end (ncclGroupStart / End).
//
// librccl.so is loaded on first use (dlopen): libxvcgpu.so itself has no RCCL
// dependency, and a process that never creates a communicator never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "xvcgpu_internal.h"

namespace {

struct Rccl {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

// loaded once; the initialisation of a function-local static is thread-safe
Rccl load_rccl() {
  Rccl r;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.dl = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.dl) break;
  }
  if (!r.dl) return r;
#define LOAD(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.dl, sym))
  LOAD(GetUniqueId, "ncclGetUniqueId");
  LOAD(CommInitRank, "ncclCommInitRank");
  LOAD(CommDestroy, "ncclCommDestroy");
  LOAD(Send, "ncclSend");
  LOAD(Recv, "ncclRecv");
  LOAD(AllReduce, "ncclAllReduce");
  LOAD(GroupStart, "ncclGroupStart");
  LOAD(GroupEnd, "ncclGroupEnd");
  LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllReduce &&
         r.GroupStart && r.GroupEnd && r.GetErrorString;
  return r;
}

Rccl &rccl() {
  static Rccl r = load_rccl();
  return r;
}

}  // namespace

struct xvcgpu_event {
  xvcgpu_ctx *ctx;
  hipEvent_t ev;
};

struct xvcgpu_comm {
  xvcgpu_ctx *ctx;
  ncclComm_t comm;
  hipStream_t stream;
  int world, rank;
  bool in_group;
};

namespace {

xvcgpu_status comm_fail(xvcgpu_ctx *ctx, xvcgpu_status st, const char *what, const char *detail) {
  if (ctx) {
    ctx->err = what;
    if (detail) {
      ctx->err += ": ";
      ctx->err += detail;
    }
  }
  return st;
}

#define CHIP_TRY(ctx, call)                                                               \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return comm_fail(ctx, XVCGPU_DEVICE_ERROR, #call, hipGetErrorString(e_));           \
  } while (0)
// a failing call between xvcgpu_comm_group_begin / _end must not leave the
// group open (everything enqueued later would silently join it): close it
// before reporting
#define NCCL_TRY(comm_or_null, ctx, call)                                                 \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess) {                                                              \
      xvcgpu_comm *c_ = (comm_or_null);                                                   \
      if (c_ && c_->in_group) {                                                           \
        c_->in_group = false;                                                             \
        rccl().GroupEnd();                                                                \
      }                                                                                   \
      return comm_fail(ctx, XVCGPU_DEVICE_ERROR, #call, rccl().GetErrorString(r_));       \
    }                                                                                     \
  } while (0)
// every entry point runs on the context's device, whatever device the calling
// thread had current (a process may hold contexts on several GPUs)
#define ON_DEVICE(ctx) CHIP_TRY(ctx, hipSetDevice((ctx)->device))

// the contiguous bytes of rows [y0, y1) of a plane, borders included
void plane_rows(const PlaneView &p, int border, int y0, int y1, uint16_t **ptr, size_t *count) {
  *ptr = p.p + (ptrdiff_t)y0 * p.stride - border;
  *count = (size_t)(y1 - y0) * (size_t)p.stride;
}

}  // namespace

extern "C" {

xvcgpu_status xvcgpu_event_create(xvcgpu_ctx *ctx, xvcgpu_event **out) {
  if (!ctx || !out) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_event *e = new (std::nothrow) xvcgpu_event();
  if (!e) return XVCGPU_OUT_OF_MEMORY;
  e->ctx = ctx;
  hipError_t r = hipSetDevice(ctx->device);
  if (r == hipSuccess) r = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
  if (r != hipSuccess) {
    delete e;
    return comm_fail(ctx, XVCGPU_DEVICE_ERROR, "hipEventCreateWithFlags", hipGetErrorString(r));
  }
  *out = e;
  return XVCGPU_OK;
}

void xvcgpu_event_destroy(xvcgpu_event *ev) {
  if (!ev) return;
  hipEventDestroy(ev->ev);
  delete ev;
}

xvcgpu_status xvcgpu_event_record(xvcgpu_ctx *ctx, xvcgpu_event *ev) {
  if (!ctx || !ev) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(ctx, hipEventRecord(ev->ev, ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_event_wait(xvcgpu_ctx *ctx, xvcgpu_event *ev) {
  if (!ctx || !ev) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev->ev, 0));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_upload_ahead(xvcgpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes,
                                  xvcgpu_event *after, xvcgpu_event *done) {
  if (!ctx || (!d_dst && bytes) || (!h_src && bytes) || !done) return XVCGPU_INVALID_ARGUMENT;
  if (!ctx->copy_stream) {
    CHIP_TRY(ctx, hipSetDevice(ctx->device));
    CHIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  }
  if (after) CHIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, after->ev, 0));
  if (bytes)
    CHIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  CHIP_TRY(ctx, hipEventRecord(done->ev, ctx->copy_stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_internal_after_wait(xvcgpu_ctx *ctx);   // xvcgpu.hip

xvcgpu_status xvcgpu_event_synchronize(xvcgpu_event *ev) {
  if (!ev) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(ev->ctx, hipEventSynchronize(ev->ev));
  return xvcgpu_internal_after_wait(ev->ctx);
}

xvcgpu_status xvcgpu_event_query(xvcgpu_event *ev, int *done) {
  if (!ev || !done) return XVCGPU_INVALID_ARGUMENT;
  const hipError_t e = hipEventQuery(ev->ev);
  if (e == hipSuccess) {
    *done = 1;
    return xvcgpu_internal_after_wait(ev->ctx);
  }
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    *done = 0;
    return XVCGPU_OK;
  }
  CHIP_TRY(ev->ctx, e);
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_unique_id(uint8_t id[XVCGPU_COMM_ID_BYTES]) {
  if (!id) return XVCGPU_INVALID_ARGUMENT;
  static_assert(XVCGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!rccl().ok) return XVCGPU_UNSUPPORTED;
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return XVCGPU_DEVICE_ERROR;
  std::memcpy(id, u.internal, XVCGPU_COMM_ID_BYTES);
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_create(xvcgpu_ctx *ctx, const uint8_t id[XVCGPU_COMM_ID_BYTES],
                                 int world, int rank, xvcgpu_comm **out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world)
    return XVCGPU_INVALID_ARGUMENT;
  if (!rccl().ok) return comm_fail(ctx, XVCGPU_UNSUPPORTED, "librccl.so not found", dlerror());
  xvcgpu_comm *c = new (std::nothrow) xvcgpu_comm();
  if (!c) return XVCGPU_OUT_OF_MEMORY;
  c->ctx = ctx;
  c->world = world;
  c->rank = rank;
  c->comm = nullptr;
  c->stream = nullptr;
  c->in_group = false;
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return comm_fail(ctx, XVCGPU_DEVICE_ERROR, "hipStreamCreateWithFlags", hipGetErrorString(e));
  }
  ncclUniqueId u;
  std::memcpy(u.internal, id, XVCGPU_COMM_ID_BYTES);
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    hipStreamDestroy(c->stream);
    delete c;
    return comm_fail(ctx, XVCGPU_DEVICE_ERROR, "ncclCommInitRank", rccl().GetErrorString(r));
  }
  *out = c;
  return XVCGPU_OK;
}

void xvcgpu_comm_destroy(xvcgpu_comm *comm) {
  if (!comm) return;
  hipStreamSynchronize(comm->stream);
  if (comm->comm) rccl().CommDestroy(comm->comm);
  hipStreamDestroy(comm->stream);
  delete comm;
}

int xvcgpu_comm_world(const xvcgpu_comm *comm) { return comm ? comm->world : 0; }
int xvcgpu_comm_rank(const xvcgpu_comm *comm) { return comm ? comm->rank : -1; }

xvcgpu_status xvcgpu_comm_wait_event(xvcgpu_comm *comm, xvcgpu_event *ev) {
  if (!comm || !ev) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(comm->ctx, hipStreamWaitEvent(comm->stream, ev->ev, 0));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_record_event(xvcgpu_comm *comm, xvcgpu_event *ev) {
  if (!comm || !ev) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(comm->ctx, hipEventRecord(ev->ev, comm->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_sync(xvcgpu_comm *comm) {
  if (!comm) return XVCGPU_INVALID_ARGUMENT;
  CHIP_TRY(comm->ctx, hipStreamSynchronize(comm->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_group_begin(xvcgpu_comm *comm) {
  if (!comm) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(nullptr, comm->ctx, rccl().GroupStart());
  comm->in_group = true;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_group_end(xvcgpu_comm *comm) {
  if (!comm) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  comm->in_group = false;
  NCCL_TRY(nullptr, comm->ctx, rccl().GroupEnd());
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_send_picture(xvcgpu_comm *comm, const xvcgpu_picture *pic, int dst) {
  if (!comm || !pic || dst < 0 || dst >= comm->world) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(comm, comm->ctx, rccl().Send(pic->base, pic->bytes, ncclUint8, dst, comm->comm, comm->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_recv_picture(xvcgpu_comm *comm, xvcgpu_picture *pic, int src) {
  if (!comm || !pic || src < 0 || src >= comm->world) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(comm, comm->ctx, rccl().Recv(pic->base, pic->bytes, ncclUint8, src, comm->comm, comm->stream));
  return XVCGPU_OK;
}

// rows [y0, y1) in luma units (even): the luma rows and the chroma rows y/2
static xvcgpu_status rows_xfer(xvcgpu_comm *comm, const xvcgpu_picture *pic, int comp_mask,
                               int y0, int y1, int peer, bool send) {
  if (!comm || !pic || peer < 0 || peer >= comm->world || y0 >= y1 || ((y0 | y1) & 1) ||
      !(comp_mask & 7))
    return XVCGPU_INVALID_ARGUMENT;
  const int bl = XVCGPU_BORDER_LUMA;
  if (y0 < -bl || y1 > pic->h + bl) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  for (int c = 0; c < 3; c++) {
    if (!(comp_mask & (1 << c))) continue;
    uint16_t *p;
    size_t n;
    plane_rows(pic->v.c[c], c ? bl / 2 : bl, c ? y0 / 2 : y0, c ? y1 / 2 : y1, &p, &n);
    if (send)
      NCCL_TRY(comm, comm->ctx, rccl().Send(p, 2 * n, ncclUint8, peer, comm->comm, comm->stream));
    else
      NCCL_TRY(comm, comm->ctx, rccl().Recv(p, 2 * n, ncclUint8, peer, comm->comm, comm->stream));
  }
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_send_rows(xvcgpu_comm *comm, const xvcgpu_picture *pic, int comp_mask,
                                    int y0, int y1, int dst) {
  return rows_xfer(comm, pic, comp_mask, y0, y1, dst, true);
}

xvcgpu_status xvcgpu_comm_recv_rows(xvcgpu_comm *comm, xvcgpu_picture *pic, int comp_mask, int y0,
                                    int y1, int src) {
  return rows_xfer(comm, pic, comp_mask, y0, y1, src, false);
}

xvcgpu_status xvcgpu_comm_send_bytes(xvcgpu_comm *comm, const void *d_src, size_t bytes, int dst) {
  if (!comm || !d_src || !bytes || dst < 0 || dst >= comm->world) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(comm, comm->ctx, rccl().Send(d_src, bytes, ncclUint8, dst, comm->comm, comm->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_recv_bytes(xvcgpu_comm *comm, void *d_dst, size_t bytes, int src) {
  if (!comm || !d_dst || !bytes || src < 0 || src >= comm->world) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(comm, comm->ctx, rccl().Recv(d_dst, bytes, ncclUint8, src, comm->comm, comm->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_comm_all_reduce_sum_u64(xvcgpu_comm *comm, uint64_t *d_values, int n) {
  if (!comm || !d_values || n < 1) return XVCGPU_INVALID_ARGUMENT;
  ON_DEVICE(comm->ctx);
  NCCL_TRY(comm, comm->ctx, rccl().AllReduce(d_values, d_values, (size_t)n, ncclUint64, ncclSum,
                                       comm->comm, comm->stream));
  return XVCGPU_OK;
}

}  // extern "C"
