// xvcgpu.hip -- C-ABI implementation of libxvcgpu.so (see include/xvcgpu.h).
// gfx950 only; no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "k_deblock.h"
#include "k_tail.h"
#include "k_me.h"
#include "k_me2.h"
#include "k_metric.h"
#include "k_misc.h"
#include "k_pad.h"
#include "k_tx.h"
#include "k_tx2.h"
#include "k_recon.h"
#include "k_bipred.h"
#include "k_stats.h"
#include "k_intra.h"
#include "k_affine_me.h"
#include "k_inter_pred.h"
#include "k_multi.h"
#include "k_rd.h"
#include "k_intra_waves.h"
#include "k_cu_state.h"
#include "k_cs_engine.h"
#include "xvcgpu_internal.h"

namespace {

xvcgpu_status fail(xvcgpu_ctx *ctx, xvcgpu_status st, const char *what,
                   hipError_t e = hipSuccess) {
  if (ctx) {
    ctx->err = what;
    if (e != hipSuccess) {
      ctx->err += ": ";
      ctx->err += hipGetErrorString(e);
    }
  }
  return st;
}

#define HIP_TRY(ctx, call)                                              \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) return fail(ctx, XVCGPU_DEVICE_ERROR, #call, e_); \
  } while (0)

#define CHECK_LAUNCH(ctx, name)                                          \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) return fail(ctx, XVCGPU_DEVICE_ERROR, name, e_); \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Blocks visited by ComparePicture on a w x h plane (sample_metric.cc:64-88).
inline int ssd_items(int w, int h) {
  const int mbx = w & ~(w - 1), mby = h & ~(h - 1);
  const int ncx = (w > 64 ? (w - 64 + 63) / 64 : 0) + (w - (w & ~63)) / mbx;
  const int ncy = (h > 64 ? (h - 64 + 63) / 64 : 0) + (h - (h & ~63)) / mby;
  return ncx * ncy;
}

// 64x64 tiles of xvcgpu_deblock_pad_ssd (k_tail.h).
inline int tail_tiles(int w, int h) { return ((w + 63) / 64) * ((h + 63) / 64); }

// Plane geometry shared by create / wrap / bytes.
struct Geometry {
  int w[3], h[3], border[3], stride[3], rows[3];
  size_t offset[3];  // byte offset of the plane's first padded row
  size_t bytes;
};

Geometry geometry(int width, int height) {
  Geometry g;
  size_t off = 0;
  for (int c = 0; c < 3; c++) {
    g.w[c] = c ? width / 2 : width;
    g.h[c] = c ? height / 2 : height;
    g.border[c] = c ? XVCGPU_BORDER_CHROMA : XVCGPU_BORDER_LUMA;
    g.stride[c] = round_up(g.w[c] + 2 * g.border[c], 64);
    g.rows[c] = g.h[c] + 2 * g.border[c];
    g.offset[c] = off;
    off += (size_t)g.stride[c] * g.rows[c] * sizeof(uint16_t);
    off = (off + 255) & ~(size_t)255;
  }
  g.bytes = off;
  return g;
}

bool valid_size(int width, int height, int bitdepth) {
  return width >= 8 && height >= 8 && (width % 8) == 0 && (height % 8) == 0 &&
         width <= 16384 && height <= 16384 && bitdepth >= 8 && bitdepth <= 12;
}

void init_views(xvcgpu_picture *p) {
  const Geometry g = geometry(p->w, p->h);
  for (int c = 0; c < 3; c++) {
    PlaneView &v = p->v.c[c];
    v.w = g.w[c];
    v.h = g.h[c];
    v.border = g.border[c];
    v.stride = g.stride[c];
    v.p = reinterpret_cast<uint16_t *>(static_cast<char *>(p->base) + g.offset[c]) +
          (size_t)g.border[c] * g.stride[c] + g.border[c];
  }
  p->v.bd = p->bd;
}

// Both residual kernels over one batch: the one-wave-per-job kernel takes the
// blocks up to 16x16, the scanning general-path kernel the rest.
#ifndef XVCGPU_INV_ONE_LAUNCH_MAX
#define XVCGPU_INV_ONE_LAUNCH_MAX 16384
#endif
template <int MODE>
void launch_residual(xvcgpu_ctx *ctx, const PicView &o, const PicView &p,
                            const PicView &r, const xvcgpu_tx_block *d_blocks, int n,
                            int16_t *d_levels, const uint32_t *d_off, int32_t *d_nnz,
                            unsigned long long *d_dist = nullptr, bool small_only = false) {
  // a decoded picture's dependency wave (a few thousand blocks, small and large mixed):
  // one launch, a workgroup per block - the two kernels below one behind the other are
  // two launches on the picture's critical path (5 + 12 us and the gap between them)
  if (MODE == TX_MODE_INV && !d_dist && !small_only && n <= XVCGPU_INV_ONE_LAUNCH_MAX &&
      ctx->inv_one_launch) {
    hipLaunchKernelGGL((residual_cu_kernel<MODE, false>), dim3(n), dim3(TX_THREADS), 0,
                       ctx->stream, o, p, r, d_blocks, n, d_levels, d_off, d_nnz,
                       ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout(), nullptr, nullptr);
    return;
  }
  const int n_wg = (n + TX2_WAVES - 1) / TX2_WAVES;
  hipLaunchKernelGGL(residual_wave_kernel<MODE>, dim3((n_wg + 7) / 8 * 8),
                     dim3(64 * TX2_WAVES), 0, ctx->stream, o, p, r, d_blocks, n,
                     d_levels, d_off, d_nnz, ctx->d_tx_tables, ctx->d_tx_tables_t,
                     xvcgpu_tx_layout(), nullptr, nullptr, d_dist);
  // (the caller knows that every block is at most 16x16: the general-path kernel
  // would find nothing to do - and still be a launch on the picture's critical path)
  if (small_only) return;
  // general path: one workgroup per block (a workgroup whose block the wave
  // kernel took retires at once); only batches far beyond a picture's worth of a
  // decoder's dependency wave take the scanning form, whose few workgroups each
  // walk their share of the large blocks one after the other (it took 0.98 ms for
  // the ~2800 blocks - 500 of them 32x32 / 64x64 - of a 1080p B picture's first
  // wave: nearly all of that picture's 1.3 ms)
  if (n <= 65536)
    hipLaunchKernelGGL(residual_per_job_kernel<MODE>, dim3(n), dim3(TX_THREADS), 0, ctx->stream,
                       o, p, r, d_blocks, n, d_levels, d_off, d_nnz, ctx->d_tx_tables,
                       xvcgpu_tx_layout(), d_dist);
  else
    hipLaunchKernelGGL(residual_kernel<MODE>, dim3((n + TX_THREADS - 1) / TX_THREADS),
                       dim3(TX_THREADS), 0, ctx->stream, o, p, r, d_blocks, n, d_levels,
                       d_off, d_nnz, ctx->d_tx_tables, xvcgpu_tx_layout(), nullptr, nullptr,
                       d_dist);
}

// TransformAndReconstruct with the RDO quantiser for the blocks that ask for it
void launch_residual_rdoq(xvcgpu_ctx *ctx, const PicView &o, const PicView &p,
                          const PicView &r, const xvcgpu_tx_block *d_blocks, int n,
                          int16_t *d_levels, const uint32_t *d_off, int32_t *d_nnz,
                          const xvcgpu_rdoq_contexts *d_ctx, const xvcgpu_rdoq_params *d_prm,
                          const xvcgpu_block_pos *d_src_pos = nullptr,
                          const xvcgpu_eval_cand *d_ecands = nullptr, int n_head = 0,
                          uint64_t *d_eout = nullptr, int strength = 0) {
  // a CU state's evaluation: its few blocks as one launch, a workgroup each
  if (n <= 64) {
    hipLaunchKernelGGL((residual_cu_kernel<TX_MODE_FULL, true>), dim3(n + (d_ecands ? n_head : 0)),
                       dim3(TX_THREADS), 0,
                       ctx->stream, o, p, r, d_blocks, n, d_levels, d_off, d_nnz,
                       ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout(), d_ctx, d_prm,
                       d_src_pos, d_ecands, n_head, d_eout, strength);
    return;
  }
  const int n_wg = (n + TX2_WAVES - 1) / TX2_WAVES;
  hipLaunchKernelGGL((residual_wave_kernel<TX_MODE_FULL, true>), dim3((n_wg + 7) / 8 * 8),
                     dim3(64 * TX2_WAVES), 0, ctx->stream, o, p, r, d_blocks, n,
                     d_levels, d_off, d_nnz, ctx->d_tx_tables, ctx->d_tx_tables_t,
                     xvcgpu_tx_layout(), d_ctx, d_prm);
  // the blocks beyond 16x16 (and the 2-wide ones): a workgroup each - the scanning
  // form runs the large blocks of its 256 descriptors one after the other, which a
  // batch of RD-search candidates (a third of them 32x32 / 64x64) turns into the
  // whole launch's duration
  if (n <= 262144)
    hipLaunchKernelGGL((residual_per_job_kernel<TX_MODE_FULL, true>), dim3(n), dim3(TX_THREADS),
                       0, ctx->stream, o, p, r, d_blocks, n, d_levels, d_off, d_nnz,
                       ctx->d_tx_tables, xvcgpu_tx_layout(), nullptr, d_ctx, d_prm);
  else
    hipLaunchKernelGGL((residual_kernel<TX_MODE_FULL, true>),
                       dim3((n + TX_THREADS - 1) / TX_THREADS), dim3(TX_THREADS), 0, ctx->stream,
                       o, p, r, d_blocks, n, d_levels, d_off, d_nnz, ctx->d_tx_tables,
                       xvcgpu_tx_layout(), d_ctx, d_prm);
}

}  // namespace

extern "C" {

const char *xvcgpu_version(void) { return "xvcgpu 0.1 gfx950"; }

xvcgpu_status xvcgpu_create(int device, xvcgpu_ctx **out) {
  if (!out) return XVCGPU_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 ||
      device >= count)
    return XVCGPU_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return XVCGPU_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return XVCGPU_NO_DEVICE;
  xvcgpu_ctx *ctx = new (std::nothrow) xvcgpu_ctx();
  if (!ctx) return XVCGPU_OUT_OF_MEMORY;
  ctx->device = device;
  ctx->stream = nullptr;
  ctx->hi_stream = nullptr;
  ctx->copy_stream = nullptr;
  ctx->own_stream = false;
  ctx->d_tx_tables = nullptr;
  ctx->d_tx_tables_t = nullptr;
  ctx->d_tz_pattern = nullptr;
  ctx->d_ssd_part = nullptr;
  ctx->d_tail_part = nullptr;
  ctx->tail_cap = 0;
  ctx->ssd_part_cap = 0;
  ctx->d_stats = nullptr;
  ctx->stats_rows_cap = 0;
  ctx->d_rdoq_lists = nullptr;
  ctx->rdoq_lists_cap = 0;
  ctx->d_crc_tables = nullptr;
  ctx->d_intra_done = nullptr;
  ctx->intra_done_cap = 0;
  ctx->intra_waves_grid = 0;
  ctx->rdoq_four_lane_only = 0;
  ctx->h_rdoq_misuse = nullptr;
  ctx->h_seg_ring = nullptr;
  ctx->seg_ring_pos = 0;
  ctx->seg_ring_half = 0;
  ctx->seg_ring_ev[0] = ctx->seg_ring_ev[1] = nullptr;
  ctx->seg_ring_used[0] = ctx->seg_ring_used[1] = false;
  {
    ctx->rdoq_qp_hint = -1;
    ctx->rdoq_classified_proved = false;
    const char *e = getenv("XVCGPU_PROVE_ZERO");   // 0 / 1; default: by batch size
    ctx->rdoq_prove_zero = (e && (e[0] == '0' || e[0] == '1') && !e[1]) ? e[0] - '0' : -1;
  }
  for (int i = 0; i < 64; i++) ctx->ev_pool[i] = nullptr;
  ctx->d_me_rot = nullptr;
  ctx->me_epoch = 0;
  {
    const char *e = getenv("XVCGPU_INV_ONE_LAUNCH");   // 0: the two-kernel form (comparisons)
    ctx->inv_one_launch = !(e && e[0] == '0' && !e[1]);
  }
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return XVCGPU_DEVICE_ERROR;
  }
  ctx->own_stream = true;
  hipEventCreate(&ctx->ev0);
  hipEventCreate(&ctx->ev1);
  hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming);
  const TxTableLayout &lay = xvcgpu_tx_layout();
  if (hipMalloc(&ctx->d_tx_tables, lay.total * sizeof(int16_t)) != hipSuccess ||
      hipMemcpy(ctx->d_tx_tables, xvcgpu_tx_host_tables(),
                lay.total * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&ctx->d_tx_tables_t, lay.total * sizeof(int16_t)) != hipSuccess ||
      hipMemcpy(ctx->d_tx_tables_t, xvcgpu_tx_host_tables_t(),
                lay.total * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess) {
    xvcgpu_destroy(ctx);
    return XVCGPU_OUT_OF_MEMORY;
  }
  if (hipMalloc(&ctx->d_me_rot, 3 * sizeof(Me2Rot)) != hipSuccess ||
      hipMemset(ctx->d_me_rot, 0x7f, 3 * sizeof(Me2Rot)) != hipSuccess) {
    xvcgpu_destroy(ctx);
    return XVCGPU_OUT_OF_MEMORY;
  }
  {
    uint32_t bits[128];
    if (hipMemcpyFromSymbol(bits, HIP_SYMBOL(kEntropyBits), sizeof(bits)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(gEntropyBits), bits, sizeof(bits)) != hipSuccess) {
      xvcgpu_destroy(ctx);
      return XVCGPU_DEVICE_ERROR;
    }
  }
  {
    TzCand pattern[TZ_MAX_CANDS];
    const int np = tz_pattern_build(pattern);
    if (np != TZ_MAX_CANDS ||
        hipMalloc(&ctx->d_tz_pattern, sizeof(pattern)) != hipSuccess ||
        hipMemcpy(ctx->d_tz_pattern, pattern, sizeof(pattern),
                  hipMemcpyHostToDevice) != hipSuccess) {
      xvcgpu_destroy(ctx);
      return XVCGPU_OUT_OF_MEMORY;
    }
  }
  // the flag the four-lane-only promise is checked with (xvcgpu_quant_rdo_set_four_lane_only):
  // allocated here, not on first use - that first use may sit inside a stream capture
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_rdoq_misuse), sizeof(int),
                    hipHostMallocMapped) != hipSuccess) {
    xvcgpu_destroy(ctx);
    return XVCGPU_OUT_OF_MEMORY;
  }
  *ctx->h_rdoq_misuse = 0;
  *out = ctx;
  return XVCGPU_OK;
}

void xvcgpu_destroy(xvcgpu_ctx *ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->d_tx_tables) hipFree(ctx->d_tx_tables);
  if (ctx->d_tx_tables_t) hipFree(ctx->d_tx_tables_t);
  if (ctx->d_tz_pattern) hipFree(ctx->d_tz_pattern);
  if (ctx->d_ssd_part) hipFree(ctx->d_ssd_part);
  if (ctx->d_tail_part) hipFree(ctx->d_tail_part);
  if (ctx->d_stats) hipFree(ctx->d_stats);
  if (ctx->d_rdoq_lists) hipFree(ctx->d_rdoq_lists);
  if (ctx->h_rdoq_misuse) hipHostFree(ctx->h_rdoq_misuse);
  if (ctx->h_seg_ring) {
    hipHostFree(ctx->h_seg_ring);
    hipEventDestroy(ctx->seg_ring_ev[0]);
    hipEventDestroy(ctx->seg_ring_ev[1]);
  }
  if (ctx->d_crc_tables) hipFree(ctx->d_crc_tables);
  if (ctx->d_intra_done) hipFree(ctx->d_intra_done);
  if (ctx->d_me_rot) hipFree(ctx->d_me_rot);
  hipEventDestroy(ctx->ev0);
  hipEventDestroy(ctx->ev1);
  hipEventDestroy(ctx->ev_sync);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->hi_stream) {
    hipStreamDestroy(ctx->hi_stream);
    hipEventDestroy(ctx->ev_hi_in);
    hipEventDestroy(ctx->ev_hi_out);
  }
  for (int i = 0; i < 64; i++)
    if (ctx->ev_pool[i]) hipEventDestroy(ctx->ev_pool[i]);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *xvcgpu_last_error(const xvcgpu_ctx *ctx) {
  return ctx ? ctx->err.c_str() : "null context";
}

xvcgpu_status xvcgpu_set_stream(xvcgpu_ctx *ctx, void *hip_stream) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  if (ctx->own_stream && ctx->stream) {
    hipStreamSynchronize(ctx->stream);
    hipStreamDestroy(ctx->stream);
  }
  // NULL is a valid hipStream_t: the device's default stream
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  ctx->own_stream = false;
  return XVCGPU_OK;
}

void *xvcgpu_get_stream(const xvcgpu_ctx *ctx) {
  return ctx ? static_cast<void *>(ctx->stream) : nullptr;
}

xvcgpu_status xvcgpu_use_own_stream(xvcgpu_ctx *ctx) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  if (ctx->own_stream) return XVCGPU_OK;
  hipStreamSynchronize(ctx->stream);
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->own_stream = true;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_use_priority_stream(xvcgpu_ctx *ctx, int high) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  int lo = 0, hi = 0;  // numerically lower = higher priority
  HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStreamSynchronize(ctx->stream);
  hipStream_t st = nullptr;
  HIP_TRY(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high ? hi : lo));
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  ctx->stream = st;
  ctx->own_stream = true;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_set_short_kernel_priority(xvcgpu_ctx *ctx, int on) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!on) {
    if (ctx->hi_stream) {
      hipStreamSynchronize(ctx->hi_stream);
      hipStreamDestroy(ctx->hi_stream);
      hipEventDestroy(ctx->ev_hi_in);
      hipEventDestroy(ctx->ev_hi_out);
      ctx->hi_stream = nullptr;
    }
    return XVCGPU_OK;
  }
  if (ctx->hi_stream) return XVCGPU_OK;
  int lo = 0, hi = 0;  // numerically lower = higher priority
  HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->hi_stream, hipStreamNonBlocking, hi));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_hi_in, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_hi_out, hipEventDisableTiming));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_wait_for(xvcgpu_ctx *ctx, xvcgpu_ctx *other) {
  if (!ctx || !other) return XVCGPU_INVALID_ARGUMENT;
  if (ctx == other || ctx->stream == other->stream) return XVCGPU_OK;
  HIP_TRY(ctx, hipEventRecord(other->ev_sync, other->stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, other->ev_sync, 0));
  return XVCGPU_OK;
}

// Every host-side wait ends here: a broken four-lane-only promise
// (xvcgpu_quant_rdo_set_four_lane_only) is reported by the first wait that follows the walk,
// whichever entry point it is.
xvcgpu_status xvcgpu_internal_after_wait(xvcgpu_ctx *ctx) {
  if (ctx && ctx->h_rdoq_misuse && *ctx->h_rdoq_misuse) {
    *ctx->h_rdoq_misuse = 0;
    return fail(ctx, XVCGPU_INVALID_ARGUMENT,
                "quant_rdo: a block outside the four-lane classes in a batch declared "
                "xvcgpu_quant_rdo_set_four_lane_only (its levels were not computed)");
  }
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_sync(xvcgpu_ctx *ctx) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return xvcgpu_internal_after_wait(ctx);
}

xvcgpu_status xvcgpu_timer_begin(xvcgpu_ctx *ctx) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_timer_end(xvcgpu_ctx *ctx, float *elapsed_ms) {
  if (!ctx || !elapsed_ms) return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
  HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_timer_mark(xvcgpu_ctx *ctx, int slot) {
  if (!ctx || slot < 0 || slot >= 64) return XVCGPU_INVALID_ARGUMENT;
  if (!ctx->ev_pool[slot]) HIP_TRY(ctx, hipEventCreate(&ctx->ev_pool[slot]));
  HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[slot], ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_timer_between(xvcgpu_ctx *ctx, int slot_a, int slot_b,
                                   float *elapsed_ms) {
  if (!ctx || !elapsed_ms || slot_a < 0 || slot_a >= 64 || slot_b < 0 || slot_b >= 64 ||
      !ctx->ev_pool[slot_a] || !ctx->ev_pool[slot_b])
    return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_pool[slot_b]));
  HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_pool[slot_a], ctx->ev_pool[slot_b]));
  return XVCGPU_OK;
}

struct xvcgpu_recording {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

xvcgpu_status xvcgpu_record_begin(xvcgpu_ctx *ctx) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  if (!ctx->own_stream)
    return fail(ctx, XVCGPU_UNSUPPORTED, "recording needs the private stream");
  HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_record_end(xvcgpu_ctx *ctx, xvcgpu_recording **out) {
  if (!ctx || !out) return XVCGPU_INVALID_ARGUMENT;
  *out = nullptr;
  hipGraph_t graph = nullptr;
  HIP_TRY(ctx, hipStreamEndCapture(ctx->stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(graph);
    return fail(ctx, XVCGPU_DEVICE_ERROR, "hipGraphInstantiate", e);
  }
  xvcgpu_recording *r = new (std::nothrow) xvcgpu_recording();
  if (!r) {
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    return XVCGPU_OUT_OF_MEMORY;
  }
  r->graph = graph;
  r->exec = exec;
  *out = r;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_replay(xvcgpu_ctx *ctx, xvcgpu_recording *rec) {
  if (!ctx || !rec) return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipGraphLaunch(rec->exec, ctx->stream));
  return XVCGPU_OK;
}

void xvcgpu_recording_destroy(xvcgpu_recording *rec) {
  if (!rec) return;
  hipGraphExecDestroy(rec->exec);
  hipGraphDestroy(rec->graph);
  delete rec;
}

xvcgpu_status xvcgpu_malloc(xvcgpu_ctx *ctx, size_t bytes, void **dev_ptr) {
  if (!ctx || !dev_ptr) return XVCGPU_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  hipError_t e = hipMalloc(dev_ptr, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "hipMalloc", e);
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_free(xvcgpu_ctx *ctx, void *dev_ptr) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  if (dev_ptr) {
    hipStreamSynchronize(ctx->stream);
    HIP_TRY(ctx, hipFree(dev_ptr));
  }
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_memcpy_h2d(xvcgpu_ctx *ctx, void *dst, const void *src,
                                size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return XVCGPU_INVALID_ARGUMENT;
  if (!bytes) return XVCGPU_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_host_alloc(xvcgpu_ctx *ctx, size_t bytes, void **host_ptr) {
  if (!ctx || !host_ptr || !bytes) return XVCGPU_INVALID_ARGUMENT;
  *host_ptr = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (hipHostMalloc(host_ptr, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    *host_ptr = nullptr;
    return XVCGPU_OUT_OF_MEMORY;
  }
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_host_free(xvcgpu_ctx *ctx, void *host_ptr) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  if (host_ptr) HIP_TRY(ctx, hipHostFree(host_ptr));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_memcpy_h2d_async(xvcgpu_ctx *ctx, void *dst, const void *src,
                                      size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return XVCGPU_INVALID_ARGUMENT;
  if (!bytes) return XVCGPU_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_memcpy_d2h_async(xvcgpu_ctx *ctx, void *dst, const void *src,
                                      size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return XVCGPU_INVALID_ARGUMENT;
  if (!bytes) return XVCGPU_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_memcpy_d2h(xvcgpu_ctx *ctx, void *dst, const void *src,
                                size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return XVCGPU_INVALID_ARGUMENT;
  if (!bytes) return XVCGPU_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return xvcgpu_internal_after_wait(ctx);
}

xvcgpu_status xvcgpu_memset(xvcgpu_ctx *ctx, void *dst, int value, size_t bytes) {
  if (!ctx || (!dst && bytes)) return XVCGPU_INVALID_ARGUMENT;
  if (!bytes) return XVCGPU_OK;
  HIP_TRY(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
  return XVCGPU_OK;
}

static xvcgpu_status ensure_ssd_part(xvcgpu_ctx *ctx, int items);
static xvcgpu_status ensure_stats(xvcgpu_ctx *ctx, int rows);
static xvcgpu_status ensure_tail(xvcgpu_ctx *ctx, int tiles);

/* ---- pictures ---- */
size_t xvcgpu_picture_bytes(int width, int height) {
  if (!valid_size(width, height, 8)) return 0;
  return geometry(width, height).bytes;
}

xvcgpu_status xvcgpu_picture_wrap(xvcgpu_ctx *ctx, int width, int height,
                                  int bitdepth, void *dev_mem, size_t bytes,
                                  xvcgpu_picture **out) {
  if (!ctx || !out) return XVCGPU_INVALID_ARGUMENT;
  *out = nullptr;
  if (!valid_size(width, height, bitdepth))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture size/bitdepth");
  if (!dev_mem || bytes < geometry(width, height).bytes ||
      (reinterpret_cast<uintptr_t>(dev_mem) & 255))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture memory");
  xvcgpu_picture *p = new (std::nothrow) xvcgpu_picture();
  if (!p) return XVCGPU_OUT_OF_MEMORY;
  p->ctx = ctx;
  p->w = width;
  p->h = height;
  p->bd = bitdepth;
  p->base = dev_mem;
  p->bytes = bytes;
  p->own = false;
  init_views(p);
  xvcgpu_status st = ensure_ssd_part(ctx, ssd_items(width, height));
  if (st == XVCGPU_OK) st = ensure_stats(ctx, 2 * height);
  if (st == XVCGPU_OK) st = ensure_tail(ctx, tail_tiles(width, height));
  if (st != XVCGPU_OK) {
    delete p;
    return st;
  }
  *out = p;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_create(xvcgpu_ctx *ctx, int width, int height,
                                    int bitdepth, xvcgpu_picture **out) {
  if (!ctx || !out) return XVCGPU_INVALID_ARGUMENT;
  *out = nullptr;
  if (!valid_size(width, height, bitdepth))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture size/bitdepth");
  const size_t bytes = geometry(width, height).bytes;
  void *mem = nullptr;
  xvcgpu_status st = xvcgpu_malloc(ctx, bytes, &mem);
  if (st != XVCGPU_OK) return st;
  hipMemsetAsync(mem, 0, bytes, ctx->stream);
  st = xvcgpu_picture_wrap(ctx, width, height, bitdepth, mem, bytes, out);
  if (st != XVCGPU_OK) {
    hipFree(mem);
    return st;
  }
  (*out)->own = true;
  return XVCGPU_OK;
}

void xvcgpu_picture_destroy(xvcgpu_picture *pic) {
  if (!pic) return;
  if (pic->own && pic->base) {
    hipStreamSynchronize(pic->ctx->stream);
    hipFree(pic->base);
  }
  delete pic;
}

static xvcgpu_status transfer(const xvcgpu_picture *pic,
                              const uint16_t *const planes[3],
                              const ptrdiff_t strides[3], int border_luma,
                              bool upload) {
  if (!pic || !planes || !strides) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_ctx *ctx = pic->ctx;
  if (border_luma < 0 || border_luma > XVCGPU_BORDER_LUMA || (border_luma & 1))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "border");
  for (int c = 0; c < 3; c++) {
    if (!planes[c]) continue;
    const PlaneView &v = pic->v.c[c];
    const int b = c ? border_luma / 2 : border_luma;
    uint16_t *dev = v.p - (ptrdiff_t)b * v.stride - b;
    // host pointer addresses sample (0,0); step back to the border start
    uint16_t *host = const_cast<uint16_t *>(planes[c]) - (ptrdiff_t)b * strides[c] - b;
    const size_t wbytes = (size_t)(v.w + 2 * b) * sizeof(uint16_t);
    const size_t rows = v.h + 2 * b;
    hipError_t e;
    if (upload)
      e = hipMemcpy2DAsync(dev, (size_t)v.stride * 2, host, (size_t)strides[c] * 2,
                           wbytes, rows, hipMemcpyHostToDevice, ctx->stream);
    else
      e = hipMemcpy2DAsync(host, (size_t)strides[c] * 2, dev, (size_t)v.stride * 2,
                           wbytes, rows, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return fail(ctx, XVCGPU_DEVICE_ERROR, "hipMemcpy2DAsync", e);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_upload(xvcgpu_picture *pic,
                                    const uint16_t *const planes[3],
                                    const ptrdiff_t strides[3]) {
  return transfer(pic, planes, strides, 0, true);
}
xvcgpu_status xvcgpu_picture_download(const xvcgpu_picture *pic,
                                      uint16_t *const planes[3],
                                      const ptrdiff_t strides[3]) {
  return transfer(pic, const_cast<const uint16_t *const *>(planes), strides, 0, false);
}
xvcgpu_status xvcgpu_picture_upload_padded(xvcgpu_picture *pic,
                                           const uint16_t *const planes[3],
                                           const ptrdiff_t strides[3],
                                           int border_luma) {
  return transfer(pic, planes, strides, border_luma, true);
}
xvcgpu_status xvcgpu_picture_download_padded(const xvcgpu_picture *pic,
                                             uint16_t *const planes[3],
                                             const ptrdiff_t strides[3],
                                             int border_luma) {
  return transfer(pic, const_cast<const uint16_t *const *>(planes), strides,
                  border_luma, false);
}

xvcgpu_status xvcgpu_picture_plane(const xvcgpu_picture *pic, int comp,
                                   void **dev_ptr, ptrdiff_t *stride) {
  if (!pic || comp < 0 || comp > 2 || !dev_ptr || !stride)
    return XVCGPU_INVALID_ARGUMENT;
  *dev_ptr = pic->v.c[comp].p;
  *stride = pic->v.c[comp].stride;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_copy(xvcgpu_ctx *ctx, xvcgpu_picture *dst,
                                  const xvcgpu_picture *src) {
  if (!ctx || !dst || !src) return XVCGPU_INVALID_ARGUMENT;
  if (dst->w != src->w || dst->h != src->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture size mismatch");
  HIP_TRY(ctx, hipMemcpyAsync(dst->base, src->base, geometry(src->w, src->h).bytes,
                              hipMemcpyDeviceToDevice, ctx->stream));
  return XVCGPU_OK;
}

/* ---- kernels ---- */
xvcgpu_status xvcgpu_pad_border(xvcgpu_ctx *ctx, xvcgpu_picture *pic) {
  if (!ctx || !pic) return XVCGPU_INVALID_ARGUMENT;
  hipLaunchKernelGGL(pad_border_kernel, dim3(pic->h + 2 * XVCGPU_BORDER_LUMA, 3),
                     dim3(128), 0, ctx->stream, pic->v);
  CHECK_LAUNCH(ctx, "pad_border");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_metric_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                  const xvcgpu_picture *b, int comp,
                                  double weight, int structural_strength,
                                  const xvcgpu_metric_cand *d_cands, int n,
                                  uint64_t *d_out) {
  if (!ctx || !a || !b || comp < 0 || comp > 2 || n < 0 || (n && (!d_cands || !d_out)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(metric_batch_kernel, dim3((n + 3) / 4), dim3(256), 0,
                     ctx->stream, a->v.c[comp], b->v.c[comp], a->bd, weight,
                     structural_strength, d_cands, n, d_out);
  CHECK_LAUNCH(ctx, "metric_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_mc_metric_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref,
                                     int structural_strength,
                                     const xvcgpu_mc_metric_cand *d_cands, int n,
                                     uint64_t *d_out) {
  if (!ctx || !orig || !ref || n < 0 || (n && (!d_cands || !d_out)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || orig->bd != ref->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_metric_kernel, dim3((n + 1) / 2), dim3(128), 0, ctx->stream,
                     orig->v.c[0], ref->v.c[0], orig->bd, structural_strength, d_cands,
                     n, d_out);
  CHECK_LAUNCH(ctx, "mc_metric_batch");
  return XVCGPU_OK;
}

static dim3 me2_grid(int n, int waves) {
  const int n_wg = (n + waves - 1) / waves;
  return dim3((n_wg + 7) / 8 * 8);
}

xvcgpu_status xvcgpu_me_search_sized(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref, int flags,
                                     const xvcgpu_me_block *d_blocks, int n,
                                     xvcgpu_me_result *d_results,
                                     int max_block_size) {
  if (!ctx || !orig || !ref || n < 0 || (n && (!d_blocks || !d_results)) ||
      !(flags & (XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL)) || max_block_size < 4 ||
      max_block_size > 64)
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || orig->bd != ref->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  // one wave per job; the LDS footprint is a compile-time function of the
  // block-size class, each class kernel skips the jobs of the other classes.
  // The phases are compile-time instances: a single-phase call gets a kernel
  // with only that phase's registers and LDS; the usual both-phases call runs
  // the fused instance (measured faster than two launches: waves in the
  // latency-bound full-pel search overlap waves in the VALU-bound sub-pel
  // search on the same SIMD).
  Me2Sched sched = {nullptr, nullptr, nullptr};
  if (flags & XVCGPU_ME_FULLPEL) {  // rotate the three records (k_me2.h)
    // the epoch only ever matters modulo 3: keep it there (no overflow after
    // 2^31 searches, slot indices always 0..2)
    const int e = ctx->me_epoch = (ctx->me_epoch + 1) % 3;
    sched.use = ctx->d_me_rot + e % 3;
    sched.record = ctx->d_me_rot + (e + 1) % 3;
    sched.clear = ctx->d_me_rot + (e + 2) % 3;
  }
  const bool lic_jobs = (flags & XVCGPU_ME_LIC_JOBS) != 0;
  // (the caller's word only counts where it can be kept: both phases, the 16 class alone)
  const bool only_sq16 = (flags & XVCGPU_ME_ONLY_SQ16) && (flags & 3) == 3 && max_block_size <= 16;
#define ME_LAUNCH_T(MS, PH, LIC)                                                        \
  hipLaunchKernelGGL((me_search_wave_kernel<MS, PH, LIC>), me2_grid(n, ME2_WAVES(MS)),  \
                     dim3(64 * ME2_WAVES(MS)), 0, ctx->stream, orig->v, ref->v,         \
                     d_blocks, n, d_results, ctx->d_tz_pattern, sched,                  \
                     max_block_size > 32 ? 64 : (max_block_size > 16 ? 32 : 16), lic_jobs)
#define ME_LAUNCH(MS, PH) ME_LAUNCH_T(MS, PH, false)
  // jobs of CUs that try local illumination compensation (XVC_ME_USE_LIC): their
  // own instances, the two phases as two launches
#define ME_LAUNCH_LIC(MS)                                        \
  do {                                                           \
    if (flags & XVCGPU_ME_FULLPEL) ME_LAUNCH_T(MS, 1, true);     \
    if (flags & XVCGPU_ME_SUBPEL) ME_LAUNCH_T(MS, 2, true);      \
  } while (0)
#define ME_LAUNCH_SQ16()                                                                \
  do {                                                                                  \
    const int ml = max_block_size > 32 ? 64 : (max_block_size > 16 ? 32 : 16);          \
    hipLaunchKernelGGL(me_search_sq16_kernel, me2_grid(n, ME2_WAVES(16)),                \
                       dim3(64 * ME2_WAVES(16)), 0, ctx->stream, orig->v, ref->v,       \
                       d_blocks, n, d_results, ctx->d_tz_pattern, sched, ml, lic_jobs,  \
                       only_sq16);                                                      \
    if (!only_sq16)                                                                     \
      hipLaunchKernelGGL(me_search_leftover_kernel, dim3((n + 63) / 64), dim3(64), 0,    \
                         ctx->stream, orig->v, ref->v, d_blocks, n, d_results,          \
                         ctx->d_tz_pattern, ml, lic_jobs);                              \
  } while (0)
#define ME_LAUNCH_CLASS(MS)                                \
  do {                                                     \
    if ((flags & 3) == 3 && MS == 16 && (flags & (XVCGPU_ME_HINT_SQ16 | XVCGPU_ME_ONLY_SQ16))) \
      ME_LAUNCH_SQ16();                                                                  \
    else if ((flags & 3) == 3) ME_LAUNCH(MS, 3);           \
    else if (flags & XVCGPU_ME_FULLPEL) ME_LAUNCH(MS, 1);  \
    else ME_LAUNCH(MS, 2);                                 \
  } while (0)
  // The larger classes run the two phases as two launches: their sub-pel
  // instance holds 23 / 76 KB of LDS per wave (one workgroup per CU), and fused
  // with it the full-pel search runs at that occupancy too (32x32: 36 + 64 us
  // apart, 120 us fused; 32x16: 38 + 112 vs 198; tools/time_me_classes.py).
#define ME_LAUNCH_SPLIT(MS)                                \
  do {                                                     \
    if (flags & XVCGPU_ME_FULLPEL) ME_LAUNCH(MS, 1);       \
    if (flags & XVCGPU_ME_SUBPEL) ME_LAUNCH(MS, 2);        \
  } while (0)
  ME_LAUNCH_CLASS(16);
  if (max_block_size > 16) ME_LAUNCH_SPLIT(32);
  if (max_block_size > 32) {
    ME_LAUNCH_SPLIT(64);
    // 64-class jobs on the packed sub-pel path: a team of four waves per job
    // (the <64, SUBPEL> wave instance above leaves those to it)
    if (flags & XVCGPU_ME_SUBPEL)
      hipLaunchKernelGGL((me_subpel_team_kernel<64, 4>), dim3((n + 7) / 8 * 8), dim3(256), 0,
                         ctx->stream, orig->v, ref->v, d_blocks, n, d_results);
  }
  if (lic_jobs) {
    ME_LAUNCH_LIC(16);
    if (max_block_size > 16) ME_LAUNCH_LIC(32);
    if (max_block_size > 32) ME_LAUNCH_LIC(64);
  }
#undef ME_LAUNCH_LIC
#undef ME_LAUNCH_SPLIT
#undef ME_LAUNCH_CLASS
#undef ME_LAUNCH
#undef ME_LAUNCH_T
  CHECK_LAUNCH(ctx, "me_search");
  return XVCGPU_OK;
}

#ifdef XVCGPU_TRACE
// developer build only (tools/trace_me.py): copy out the ME phase timestamps
xvcgpu_status xvcgpu_debug_me_trace(unsigned long long *out, int n_jobs) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me2_trace),
                             sizeof(unsigned long long) * 24 * (size_t)n_jobs) == hipSuccess
             ? XVCGPU_OK
             : XVCGPU_DEVICE_ERROR;
}
#endif

#ifdef XVCGPU_TRACE
// developer build only (tools/trace_rdoq.py): the walk's section clock readings
xvcgpu_status xvcgpu_debug_rdoq_trace(unsigned long long *out, int n_rows) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rq_trace),
                             sizeof(unsigned long long) * 16 * (size_t)n_rows) == hipSuccess
             ? XVCGPU_OK
             : XVCGPU_DEVICE_ERROR;
}
#endif

xvcgpu_status xvcgpu_me_search(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                               const xvcgpu_picture *ref, int flags,
                               const xvcgpu_me_block *d_blocks, int n,
                               xvcgpu_me_result *d_results) {
  return xvcgpu_me_search_sized(ctx, orig, ref, flags, d_blocks, n, d_results, 64);
}

xvcgpu_status xvcgpu_mc_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                              xvcgpu_picture *pred,
                              const xvcgpu_mc_block *d_blocks, int n) {
  if (!ctx || !ref || !pred || n < 0 || (n && !d_blocks))
    return XVCGPU_INVALID_ARGUMENT;
  if (pred->w != ref->w || pred->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_batch_kernel, dim3(n), dim3(256), 0, ctx->stream, ref->v,
                     pred->v, d_blocks, n);
  CHECK_LAUNCH(ctx, "mc_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_mc_lic_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                  const xvcgpu_picture *rec, xvcgpu_picture *pred,
                                  const xvcgpu_mc_lic_block *d_blocks, int n) {
  if (!ctx || !ref || !rec || !pred || n < 0 || (n && !d_blocks))
    return XVCGPU_INVALID_ARGUMENT;
  if (pred->w != ref->w || pred->h != ref->h || rec->w != ref->w || rec->h != ref->h ||
      rec->bd != ref->bd || pred->bd != ref->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_lic_kernel, dim3(n), dim3(256), 0, ctx->stream, ref->v, rec->v,
                     pred->v, d_blocks, n);
  CHECK_LAUNCH(ctx, "mc_lic_batch");
  return XVCGPU_OK;
}

static xvcgpu_status inter_pred_launch(xvcgpu_ctx *ctx, const xvcgpu_picture *const *refs,
                                       int n_refs, const xvcgpu_picture *rec,
                                       xvcgpu_picture *pred, const xvcgpu_inter_block *d_blocks,
                                       const xvcgpu_block_pos *d_dst, int n) {
  if (!ctx || !refs || n_refs < 1 || n_refs > XVC_MAX_REF_SLOTS || !rec || !pred || n < 0 ||
      (n && !d_blocks))
    return XVCGPU_INVALID_ARGUMENT;
  // a scratch destination has its own size; the pictures of the sequence agree
  if (rec->bd != pred->bd || (!d_dst && (rec->w != pred->w || rec->h != pred->h)))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  RefTable t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < n_refs; i++) {
    if (!refs[i]) return XVCGPU_INVALID_ARGUMENT;
    if (refs[i]->w != rec->w || refs[i]->h != rec->h || refs[i]->bd != rec->bd)
      return fail(ctx, XVCGPU_INVALID_ARGUMENT, "reference picture mismatch");
    t.pic[i] = refs[i]->v;
  }
  t.n = n_refs;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(inter_pred_kernel, dim3(n), dim3(256), 0, ctx->stream, t, rec->v,
                     pred->v, d_blocks, n, d_dst, rec->w, rec->h);
  CHECK_LAUNCH(ctx, "inter_pred_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_inter_pred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *const *refs,
                                      int n_refs, const xvcgpu_picture *rec,
                                      xvcgpu_picture *pred,
                                      const xvcgpu_inter_block *d_blocks, int n) {
  return inter_pred_launch(ctx, refs, n_refs, rec, pred, d_blocks, nullptr, n);
}

xvcgpu_status xvcgpu_inter_pred_batch_to(xvcgpu_ctx *ctx, const xvcgpu_picture *const *refs,
                                         int n_refs, const xvcgpu_picture *rec,
                                         xvcgpu_picture *scratch,
                                         const xvcgpu_inter_block *d_blocks,
                                         const xvcgpu_block_pos *d_dst, int n) {
  if (n && !d_dst) return XVCGPU_INVALID_ARGUMENT;
  return inter_pred_launch(ctx, refs, n_refs, rec, scratch, d_blocks, d_dst, n);
}

xvcgpu_status xvcgpu_copy_blocks(xvcgpu_ctx *ctx, const xvcgpu_picture *src,
                                 xvcgpu_picture *dst, const xvcgpu_copy_block *d_blocks,
                                 int n) {
  if (!ctx || !src || !dst || n < 0 || (n && !d_blocks)) return XVCGPU_INVALID_ARGUMENT;
  if (src->bd != dst->bd) return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(copy_blocks_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, src->v,
                     dst->v, d_blocks, n);
  CHECK_LAUNCH(ctx, "copy_blocks");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_mc_affine_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                     xvcgpu_picture *pred,
                                     const xvcgpu_mc_affine_block *d_blocks, int n) {
  if (!ctx || !ref || !pred || n < 0 || (n && !d_blocks)) return XVCGPU_INVALID_ARGUMENT;
  if (pred->w != ref->w || pred->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_affine_kernel, dim3(n), dim3(256), 0, ctx->stream, ref->v,
                     pred->v, d_blocks, n);
  CHECK_LAUNCH(ctx, "mc_affine_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_mc_bipred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *ref0,
                                     const xvcgpu_picture *ref1, xvcgpu_picture *pred,
                                     const xvcgpu_mc_bi_block *d_blocks, int n) {
  if (!ctx || !ref0 || !ref1 || !pred || n < 0 || (n && !d_blocks))
    return XVCGPU_INVALID_ARGUMENT;
  if (pred->w != ref0->w || pred->h != ref0->h || ref1->w != ref0->w ||
      ref1->h != ref0->h || ref1->bd != ref0->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_bipred_kernel, dim3(n), dim3(256), 0, ctx->stream, ref0->v,
                     ref1->v, pred->v, d_blocks, n);
  CHECK_LAUNCH(ctx, "mc_bipred_batch");
  return XVCGPU_OK;
}

static xvcgpu_status bipred_search_launch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *ref_other,
                                          const xvcgpu_picture *ref_search,
                                          const xvcgpu_picture *rec,
                                          const xvcgpu_bi_block *d_jobs,
                                          const xvcgpu_mc_lic_block *d_nb, int n,
                                          xvcgpu_me_result *d_results, int max_block_size) {
  if (!ctx || !orig || !ref_other || !ref_search || n < 0 ||
      (n && (!d_jobs || !d_results)) || max_block_size < 4 || max_block_size > 64 ||
      ((rec != nullptr) != (d_nb != nullptr)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref_other->w || orig->h != ref_other->h ||
      orig->bd != ref_other->bd || orig->w != ref_search->w ||
      orig->h != ref_search->h || orig->bd != ref_search->bd ||
      (rec && (rec->w != orig->w || rec->h != orig->h || rec->bd != orig->bd)))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  const dim3 grid((n + 7) / 8 * 8);
  const int bi_max = max_block_size > 32 ? 64 : (max_block_size > 16 ? 32 : 16);
#define BI_LAUNCH(MS)                                                                        \
  do {                                                                                       \
    if (rec)                                                                                 \
      hipLaunchKernelGGL((bipred_search_kernel<MS, true>), grid, dim3(64 * BI_WAVES(MS)), 0, \
                         ctx->stream, orig->v.c[0], ref_other->v.c[0], ref_search->v.c[0],   \
                         orig->bd, d_jobs, n, d_results, bi_max, rec->v.c[0], d_nb);         \
    else                                                                                     \
      hipLaunchKernelGGL((bipred_search_kernel<MS, false>), grid, dim3(64 * BI_WAVES(MS)), 0,\
                         ctx->stream, orig->v.c[0], ref_other->v.c[0], ref_search->v.c[0],   \
                         orig->bd, d_jobs, n, d_results, bi_max, PlaneView(), nullptr);      \
  } while (0)
  BI_LAUNCH(16);
  if (max_block_size > 16) BI_LAUNCH(32);
  if (max_block_size > 32) BI_LAUNCH(64);
#undef BI_LAUNCH
  CHECK_LAUNCH(ctx, "bipred_search");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_bipred_search(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *ref_other,
                                   const xvcgpu_picture *ref_search,
                                   const xvcgpu_bi_block *d_jobs, int n,
                                   xvcgpu_me_result *d_results, int max_block_size) {
  return bipred_search_launch(ctx, orig, ref_other, ref_search, nullptr, d_jobs, nullptr, n,
                              d_results, max_block_size);
}

xvcgpu_status xvcgpu_bipred_search_lic(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                       const xvcgpu_picture *ref_other,
                                       const xvcgpu_picture *ref_search,
                                       const xvcgpu_picture *rec,
                                       const xvcgpu_bi_block *d_jobs,
                                       const xvcgpu_mc_lic_block *d_neighbours, int n,
                                       xvcgpu_me_result *d_results, int max_block_size) {
  if (!rec || !d_neighbours) return XVCGPU_INVALID_ARGUMENT;
  return bipred_search_launch(ctx, orig, ref_other, ref_search, rec, d_jobs, d_neighbours, n,
                              d_results, max_block_size);
}

xvcgpu_status xvcgpu_mc_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *ref,
                                xvcgpu_picture *pred,
                                const xvcgpu_me_block *d_blocks,
                                const xvcgpu_me_result *d_results, int n) {
  if (!ctx || !ref || !pred || n < 0 || (n && (!d_blocks || !d_results)))
    return XVCGPU_INVALID_ARGUMENT;
  if (pred->w != ref->w || pred->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_from_me_kernel, dim3(n, 3), dim3(256), 0, ctx->stream,
                     ref->v, pred->v, d_blocks, d_results, n);
  CHECK_LAUNCH(ctx, "mc_from_me");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_cu_info_from_me(xvcgpu_ctx *ctx,
                                     const xvcgpu_me_block *d_blocks,
                                     const xvcgpu_me_result *d_results,
                                     const int32_t *d_nnz,
                                     const int32_t *d_luma_tx_index, int n,
                                     int qp_y, int qp_c, int ref_poc,
                                     xvcgpu_cu_info *d_cus) {
  if (!ctx || n < 0 || (n && (!d_blocks || !d_results || !d_nnz || !d_cus)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(cu_info_from_me_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     ctx->stream, d_blocks, d_results, d_nnz, d_luma_tx_index, n,
                     qp_y, qp_c, ref_poc, d_cus);
  CHECK_LAUNCH(ctx, "cu_info_from_me");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_recon_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *ref, xvcgpu_picture *rec,
                                   const xvcgpu_me_block *d_blocks,
                                   const xvcgpu_me_result *d_results, int n,
                                   int qp_y, int qp_c, int intra_pic, int ref_poc,
                                   int32_t *d_nnz, xvcgpu_cu_info *d_cus) {
  if (!ctx || !orig || !ref || !rec || n < 0 || (n && (!d_blocks || !d_results)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || rec->w != ref->w || rec->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  const int n_wg = (2 * n + 3) / 4;
  hipLaunchKernelGGL(recon_from_me_kernel<false>, dim3((n_wg + 7) / 8 * 8), dim3(256), 0,
                     ctx->stream, orig->v, ref->v, rec->v, d_blocks, d_results, n, qp_y,
                     qp_c, intra_pic, ref_poc, d_nnz, d_cus, ctx->d_tx_tables,
                     ctx->d_tx_tables_t, xvcgpu_tx_layout(), nullptr, nullptr);
  CHECK_LAUNCH(ctx, "recon_from_me");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_fwd_from_me(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                 const xvcgpu_picture *ref, xvcgpu_picture *pred,
                                 const xvcgpu_me_block *d_blocks,
                                 const xvcgpu_me_result *d_results, int n, int16_t *d_coeffs,
                                 const uint32_t *d_coeff_offsets) {
  if (!ctx || !orig || !ref || !pred || n < 0 ||
      (n && (!d_blocks || !d_results || !d_coeffs || !d_coeff_offsets)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || pred->w != ref->w || pred->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  const int n_wg = (2 * n + 3) / 4;
  hipLaunchKernelGGL((recon_from_me_kernel<false, true>), dim3((n_wg + 7) / 8 * 8), dim3(256), 0,
                     ctx->stream, orig->v, ref->v, pred->v, d_blocks, d_results, n, 0, 0, 0, 0,
                     nullptr, nullptr, ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout(),
                     nullptr, nullptr, d_coeffs, d_coeff_offsets);
  CHECK_LAUNCH(ctx, "fwd_from_me");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_recon_from_me_rdoq(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                        const xvcgpu_picture *ref, xvcgpu_picture *rec,
                                        const xvcgpu_me_block *d_blocks,
                                        const xvcgpu_me_result *d_results, int n,
                                        int qp_y, int qp_c, int tx_flags, int ref_poc,
                                        int32_t *d_nnz, xvcgpu_cu_info *d_cus,
                                        const xvcgpu_rdoq_contexts *d_contexts,
                                        const xvcgpu_rdoq_params *d_params) {
  if (!ctx || !orig || !ref || !rec || n < 0 ||
      (n && (!d_blocks || !d_results || !d_contexts || !d_params)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || rec->w != ref->w || rec->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  const int n_wg = (2 * n + 3) / 4;
  hipLaunchKernelGGL(recon_from_me_kernel<true>, dim3((n_wg + 7) / 8 * 8), dim3(256), 0,
                     ctx->stream, orig->v, ref->v, rec->v, d_blocks, d_results, n, qp_y,
                     qp_c, tx_flags | XVC_TXF_RDOQ, ref_poc, d_nnz, d_cus, ctx->d_tx_tables,
                     ctx->d_tx_tables_t, xvcgpu_tx_layout(), d_contexts, d_params);
  CHECK_LAUNCH(ctx, "recon_from_me_rdoq");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_residual_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                    const xvcgpu_picture *pred,
                                    xvcgpu_picture *rec,
                                    const xvcgpu_tx_block *d_blocks, int n,
                                    int16_t *d_levels,
                                    const uint32_t *d_level_offsets,
                                    int32_t *d_nnz) {
  if (!ctx || !orig || !pred || !rec || n < 0 || (n && !d_blocks))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  launch_residual<TX_MODE_FULL>(ctx, orig->v, pred->v, rec->v, d_blocks, n, d_levels,
                                d_level_offsets, d_nnz);
  CHECK_LAUNCH(ctx, "residual_batch");
  return XVCGPU_OK;
}

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
                                            uint64_t *d_eval_out) {
  if (!ctx || !orig || !pred || !rec || n < 0 || n_eval_head < 0 || n_eval_head > 64 ||
      (n && (!d_blocks || !d_contexts || !d_params || !d_src_pos)) ||
      ((d_eval_cands != nullptr) != (d_eval_out != nullptr)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->bd != pred->bd || rec->bd != pred->bd || rec->w != pred->w || rec->h != pred->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n > 64) return fail(ctx, XVCGPU_UNSUPPORTED, "residual_rdoq_batch_at: a CU state's blocks (<= 64)");
  if (n == 0) return XVCGPU_OK;
  launch_residual_rdoq(ctx, orig->v, pred->v, rec->v, d_blocks, n, d_levels, d_level_offsets,
                       d_nnz, d_contexts, d_params, d_src_pos, d_eval_cands, n_eval_head,
                       d_eval_out, structural_strength);
  CHECK_LAUNCH(ctx, "residual_rdoq_batch_at");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_residual_rdoq_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                         const xvcgpu_picture *pred, xvcgpu_picture *rec,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         int16_t *d_levels, const uint32_t *d_level_offsets,
                                         int32_t *d_nnz,
                                         const xvcgpu_rdoq_contexts *d_contexts,
                                         const xvcgpu_rdoq_params *d_params) {
  if (!ctx || !orig || !pred || !rec || n < 0 || (n && (!d_blocks || !d_contexts || !d_params)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != pred->w || orig->h != pred->h || rec->w != pred->w || rec->h != pred->h ||
      orig->bd != pred->bd || rec->bd != pred->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  launch_residual_rdoq(ctx, orig->v, pred->v, rec->v, d_blocks, n, d_levels, d_level_offsets,
                       d_nnz, d_contexts, d_params);
  CHECK_LAUNCH(ctx, "residual_rdoq_batch");
  return XVCGPU_OK;
}

// workgroups per class of quant_rdo_packed_kernel (k_rdoq.h)
#ifndef RDOQ_GRID16
#define RDOQ_GRID16 8192   // blocks of up to sixteen sub-blocks: one per wave
#endif
#ifndef RDOQ_GRID4
#define RDOQ_GRID4 2048    // blocks of up to four sub-blocks: four per wave
#endif
#ifndef RDOQ_GRID64
#define RDOQ_GRID64 512
#endif

static xvcgpu_status ensure_rdoq_scratch(xvcgpu_ctx *ctx, int n, size_t n_coeffs) {
  // scratch: class lists (3 x n) + counters + the per-block classes
  if (n > ctx->rdoq_lists_cap) {
    if (ctx->d_rdoq_lists) {
      hipStreamSynchronize(ctx->stream);
      hipFree(ctx->d_rdoq_lists);
      ctx->d_rdoq_lists = nullptr;
      ctx->rdoq_lists_cap = 0;
    }
    const int cap = n + n / 4;
    // count[4], three lists and the classes (cap bytes = cap / 4 ints), then the
    // compaction's per-chunk counts
    if (hipMalloc(&ctx->d_rdoq_lists,
                  sizeof(int) * (4 * (size_t)cap + 4 + 4 * ((size_t)cap / RDOQ_CHUNK + 2))) !=
        hipSuccess)
      return fail(ctx, XVCGPU_OUT_OF_MEMORY, "rdoq lists");
    ctx->rdoq_lists_cap = cap;
  }
  (void)n_coeffs;  // the per-coefficient records live in LDS
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_quant_rdo_class_counts(xvcgpu_ctx *ctx, int32_t out[3]) {
  if (!ctx || !out) return XVCGPU_INVALID_ARGUMENT;
  out[0] = out[1] = out[2] = 0;
  if (!ctx->d_rdoq_lists) return XVCGPU_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_rdoq_lists, 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_quant_rdo_set_prove_zero(xvcgpu_ctx *ctx, int mode) {
  if (!ctx || mode < -1 || mode > 1) return XVCGPU_INVALID_ARGUMENT;
  ctx->rdoq_prove_zero = mode;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_quant_rdo_set_four_lane_only(xvcgpu_ctx *ctx, int on) {
  if (!ctx) return XVCGPU_INVALID_ARGUMENT;
  ctx->rdoq_four_lane_only = on ? 1 : 0;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_quant_rdo_reserve(xvcgpu_ctx *ctx, int n, size_t n_coeffs) {
  if (!ctx || n < 0) return XVCGPU_INVALID_ARGUMENT;
  return ensure_rdoq_scratch(ctx, n, n_coeffs);
}

static RdoqLists rdoq_lists_of(xvcgpu_ctx *ctx) {
  const int cap = ctx->rdoq_lists_cap;
  RdoqLists l;
  l.count = ctx->d_rdoq_lists;
  for (int c = 0; c < 3; c++) l.list[c] = ctx->d_rdoq_lists + 4 + (size_t)c * cap;
  l.cls = reinterpret_cast<signed char *>(ctx->d_rdoq_lists + 4 + 3 * (size_t)cap);
  l.part = ctx->d_rdoq_lists + 4 + 4 * (size_t)cap;
  return l;
}

// classified: the blocks' classes are already in the context's RdoqLists::cls
// (written by the forward transform of xvcgpu_frame_pass, FwdClassify)
static xvcgpu_status quant_rdo_launch(xvcgpu_ctx *ctx, int bitdepth,
                                      const xvcgpu_tx_block *d_blocks, int n,
                                      const int16_t *d_coeffs, const uint32_t *d_offsets,
                                      size_t n_coeffs, int16_t *d_levels, int32_t *d_nnz,
                                      const xvcgpu_rdoq_contexts *d_contexts,
                                      const xvcgpu_rdoq_params *d_params, bool classified,
                                      xvcgpu_cu_info *d_cu_patch = nullptr) {
  if (!ctx || n < 0 || bitdepth < 8 || bitdepth > 12 ||
      (n && (!d_blocks || !d_coeffs || !d_offsets || !d_levels || !d_contexts || !d_params ||
             !n_coeffs)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  {
    const xvcgpu_status st_ = ensure_rdoq_scratch(ctx, n, n_coeffs);
    if (st_ != XVCGPU_OK) return st_;
  }
  const RdoqLists l = rdoq_lists_of(ctx);
  if (!classified)
    hipLaunchKernelGGL(rdoq_classify_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream,
                       bitdepth, d_blocks, n, d_coeffs, d_offsets, d_levels, d_nnz, l);
  // the blocks the walk is bound to return 0 for leave the lists here (k_rdoq.h)
  const int qp_hint = classified ? ctx->rdoq_qp_hint : -1;
  // (the classes a proving forward call left behind stay proved until the next one)
  const bool proved_already = classified && ctx->rdoq_classified_proved;
  if (!proved_already &&
      (ctx->rdoq_prove_zero > 0 ||
       (ctx->rdoq_prove_zero < 0 && n >= XVCGPU_PROVE_ZERO_AUTO_BLOCKS &&
        (qp_hint < 0 || qp_hint >= XVCGPU_PROVE_ZERO_AUTO_QP))))
    hipLaunchKernelGGL(rdoq_prove_zero_kernel, dim3((n + RQ_PROVE_BLOCKS - 1) / RQ_PROVE_BLOCKS),
                       dim3(256), 0, ctx->stream,
                       bitdepth, d_blocks, n, d_coeffs, d_offsets, d_levels, d_nnz, d_contexts,
                       d_params, l);
  {
    const int chunks = (n + RDOQ_CHUNK - 1) / RDOQ_CHUNK;
    hipLaunchKernelGGL(rdoq_count_kernel, dim3(chunks), dim3(1024), 0, ctx->stream, n, l);
    hipLaunchKernelGGL(rdoq_scatter_kernel, dim3(chunks), dim3(1024), 0, ctx->stream, n, l);
  }
  // the class sizes are only known on the device: a bounded number of workgroups
  // per class that walk their list (k_rdoq.h)
  static const int grid16 = [] {
    const char *e = getenv("XVCGPU_RDOQ_GRID16");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : RDOQ_GRID16;
  }();
  const int g16 = std::min(n, grid16), g4 = std::min((n + 3) / 4, RDOQ_GRID4),
            g64 = std::min(n, RDOQ_GRID64);
  // (the general class's launch holds 255 vector registers a wave: even with an empty list
  // it waits for room beside other streams' kernels - 190 us in flight at 2160p; a caller
  // that knows its blocks says so and the launch is not made)
  const bool four_only = ctx->rdoq_four_lane_only != 0;
  hipLaunchKernelGGL(quant_rdo_packed4_kernel, dim3(g16 + g4), dim3(64), 0, ctx->stream,
                     bitdepth, d_blocks, l, g16, d_coeffs, d_offsets, d_levels, d_nnz,
                     d_contexts, d_params, d_cu_patch, four_only ? ctx->h_rdoq_misuse : nullptr);
  if (!four_only)
    hipLaunchKernelGGL(quant_rdo_packed_kernel, dim3(g64), dim3(64), 0, ctx->stream,
                       bitdepth, d_blocks, l, d_coeffs, d_offsets, d_levels, d_nnz,
                       d_contexts, d_params, d_cu_patch);
  CHECK_LAUNCH(ctx, "quant_rdo_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_quant_rdo_batch(xvcgpu_ctx *ctx, int bitdepth,
                                     const xvcgpu_tx_block *d_blocks, int n,
                                     const int16_t *d_coeffs, const uint32_t *d_offsets,
                                     size_t n_coeffs, int16_t *d_levels, int32_t *d_nnz,
                                     const xvcgpu_rdoq_contexts *d_contexts,
                                     const xvcgpu_rdoq_params *d_params) {
  return quant_rdo_launch(ctx, bitdepth, d_blocks, n, d_coeffs, d_offsets, n_coeffs, d_levels,
                          d_nnz, d_contexts, d_params, false);
}

xvcgpu_status xvcgpu_quant_rdo_classified_batch(xvcgpu_ctx *ctx, int bitdepth,
                                                const xvcgpu_tx_block *d_blocks, int n,
                                                const int16_t *d_coeffs,
                                                const uint32_t *d_offsets, size_t n_coeffs,
                                                int16_t *d_levels, int32_t *d_nnz,
                                                const xvcgpu_rdoq_contexts *d_contexts,
                                                const xvcgpu_rdoq_params *d_params,
                                                xvcgpu_cu_info *d_cus) {
  return quant_rdo_launch(ctx, bitdepth, d_blocks, n, d_coeffs, d_offsets, n_coeffs, d_levels,
                          d_nnz, d_contexts, d_params, true, d_cus);
}

xvcgpu_status xvcgpu_fwd_from_me_classify(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *ref, xvcgpu_picture *pred,
                                          const xvcgpu_me_block *d_blocks,
                                          const xvcgpu_me_result *d_results, int n, int qp_y,
                                          int qp_c, int ref_poc, int16_t *d_coeffs,
                                          const uint32_t *d_coeff_offsets, size_t n_coeffs,
                                          int16_t *d_levels, int32_t *d_nnz,
                                          xvcgpu_cu_info *d_cus) {
  return xvcgpu_fwd_from_me_classify_prove(ctx, orig, ref, pred, d_blocks, d_results, n, qp_y, qp_c,
                                           ref_poc, d_coeffs, d_coeff_offsets, n_coeffs, d_levels,
                                           d_nnz, d_cus, nullptr, nullptr);
}

xvcgpu_status xvcgpu_fwd_from_me_classify_prove(
    xvcgpu_ctx *ctx, const xvcgpu_picture *orig, const xvcgpu_picture *ref, xvcgpu_picture *pred,
    const xvcgpu_me_block *d_blocks, const xvcgpu_me_result *d_results, int n, int qp_y, int qp_c,
    int ref_poc, int16_t *d_coeffs, const uint32_t *d_coeff_offsets, size_t n_coeffs,
    int16_t *d_levels, int32_t *d_nnz, xvcgpu_cu_info *d_cus,
    const xvcgpu_rdoq_contexts *d_contexts, const xvcgpu_rdoq_params *d_params) {
  if ((d_contexts == nullptr) != (d_params == nullptr)) return XVCGPU_INVALID_ARGUMENT;
  if (!ctx || !orig || !ref || !pred || n < 0 ||
      (n && (!d_blocks || !d_results || !d_coeffs || !d_coeff_offsets || !d_levels || !d_nnz)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != ref->w || orig->h != ref->h || pred->w != ref->w || pred->h != ref->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  {
    const xvcgpu_status st = ensure_rdoq_scratch(ctx, 3 * n, n_coeffs);
    if (st != XVCGPU_OK) return st;
  }
  ctx->rdoq_qp_hint = qp_y;
  FwdClassify fc;
  fc.cls = rdoq_lists_of(ctx).cls;
  fc.levels = d_levels;
  fc.nnz = d_nnz;
  // the all-zero proof where the coefficients are at hand, unless it is switched off;
  // the quantiser's own launch of it is then not needed
  const bool prove = d_contexts && ctx->rdoq_prove_zero != 0;
  fc.rq_ctx = prove ? d_contexts : nullptr;
  fc.rq_prm = prove ? d_params : nullptr;
  fc.pv = nullptr;
  ctx->rdoq_classified_proved = prove;
  const int n_wg = (2 * n + 3) / 4;
  hipLaunchKernelGGL((recon_from_me_kernel<false, true>), dim3((n_wg + 7) / 8 * 8), dim3(256), 0,
                     ctx->stream, orig->v, ref->v, pred->v, d_blocks, d_results, n, qp_y, qp_c, 0,
                     ref_poc, nullptr, d_cus, ctx->d_tx_tables, ctx->d_tx_tables_t,
                     xvcgpu_tx_layout(), nullptr, nullptr, d_coeffs, d_coeff_offsets, fc);
  CHECK_LAUNCH(ctx, "fwd_from_me_classify");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_fwd_transform_batch(xvcgpu_ctx *ctx,
                                         const xvcgpu_picture *orig,
                                         const xvcgpu_picture *pred,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         int16_t *d_coeffs,
                                         const uint32_t *d_coeff_offsets) {
  if (!ctx || !orig || !pred || n < 0 ||
      (n && (!d_blocks || !d_coeffs || !d_coeff_offsets)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  launch_residual<TX_MODE_FWD>(ctx, orig->v, pred->v, pred->v, d_blocks, n, d_coeffs,
                               d_coeff_offsets, (int32_t *)nullptr);
  CHECK_LAUNCH(ctx, "fwd_transform_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_inv_transform_batch(xvcgpu_ctx *ctx,
                                         const xvcgpu_picture *pred,
                                         xvcgpu_picture *rec,
                                         const xvcgpu_tx_block *d_blocks, int n,
                                         const int16_t *d_levels,
                                         const uint32_t *d_level_offsets,
                                         const int32_t *d_nnz) {
  if (!ctx || !pred || !rec || n < 0 ||
      (n && (!d_blocks || !d_levels || !d_level_offsets || !d_nnz)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  launch_residual<TX_MODE_INV>(ctx, pred->v, pred->v, rec->v, d_blocks, n,
                               const_cast<int16_t *>(d_levels), d_level_offsets,
                               const_cast<int32_t *>(d_nnz));
  CHECK_LAUNCH(ctx, "inv_transform_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_inv_transform_cu_order(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                            const xvcgpu_tx_block *d_blocks, int n_cus,
                                            const int16_t *d_levels,
                                            const uint32_t *d_level_offsets,
                                            const int32_t *d_nnz) {
  if (!ctx || !rec || n_cus < 0 ||
      (n_cus && (!d_blocks || !d_levels || !d_level_offsets || !d_nnz)))
    return XVCGPU_INVALID_ARGUMENT;
  if (n_cus == 0) return XVCGPU_OK;
  const int n_wg = (2 * n_cus + TX2_WAVES - 1) / TX2_WAVES;
  hipLaunchKernelGGL(inv_cu_pairs_kernel, dim3((n_wg + 7) / 8 * 8), dim3(64 * TX2_WAVES), 0,
                     ctx->stream, rec->v, d_blocks, n_cus, const_cast<int16_t *>(d_levels),
                     d_level_offsets, const_cast<int32_t *>(d_nnz), ctx->d_tx_tables,
                     ctx->d_tx_tables_t, xvcgpu_tx_layout());
  CHECK_LAUNCH(ctx, "inv_transform_cu_order");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_inv_transform_dist_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                              const xvcgpu_picture *pred, xvcgpu_picture *rec,
                                              const xvcgpu_tx_block *d_blocks, int n,
                                              const int16_t *d_levels,
                                              const uint32_t *d_level_offsets,
                                              const int32_t *d_nnz, uint64_t *d_dist) {
  if (!ctx || !orig || !pred || !rec || n < 0 ||
      (n && (!d_blocks || !d_levels || !d_level_offsets || !d_nnz || !d_dist)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->w != pred->w || orig->h != pred->h || orig->bd != pred->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  launch_residual<TX_MODE_INV>(ctx, orig->v, pred->v, rec->v, d_blocks, n,
                               const_cast<int16_t *>(d_levels), d_level_offsets,
                               const_cast<int32_t *>(d_nnz),
                               reinterpret_cast<unsigned long long *>(d_dist));
  CHECK_LAUNCH(ctx, "inv_transform_dist_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_tx_eval_batch(xvcgpu_ctx *ctx, const xvcgpu_tx_eval_job *d_jobs, int n,
                                   const xvcgpu_tx_eval_alt *d_alts,
                                   xvcgpu_tx_eval_result *d_out) {
  if (!ctx || n < 0 || (n && (!d_jobs || !d_alts || !d_out))) return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(tx_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, n,
                     d_alts, d_out);
  CHECK_LAUNCH(ctx, "tx_eval_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_root_cbf_batch(xvcgpu_ctx *ctx, const xvcgpu_root_cbf_job *d_jobs, int n,
                                    xvcgpu_root_cbf_result *d_out) {
  if (!ctx || n < 0 || (n && (!d_jobs || !d_out))) return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(root_cbf_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, n,
                     d_out);
  CHECK_LAUNCH(ctx, "root_cbf_batch");
  return XVCGPU_OK;
}

static xvcgpu_status deblock_rows_masked(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                         const xvcgpu_cu_info *d_cus, int n_cus,
                                         const int32_t *d_cu_map, int map_stride,
                                         int pic_is_bipred, int beta_offset,
                                         int tc_offset, int subblock_size, int pass,
                                         int y_begin, int y_end, int comp_mask) {
  if (!ctx || !rec || !d_cus || n_cus <= 0 || !d_cu_map ||
      map_stride < (rec->w + 3) / 4 || (subblock_size != 4 && subblock_size != 8) ||
      (pass != 0 && pass != 1) || y_begin < 0 || (y_begin % subblock_size) != 0)
    return XVCGPU_INVALID_ARGUMENT;
  if (y_end > rec->h) y_end = rec->h;
  if (y_end <= y_begin) return XVCGPU_OK;
  DbParams d;
  d.bd = rec->bd;
  d.pic_w = rec->w;
  d.pic_h = rec->h;
  d.bipred = pic_is_bipred;
  d.beta_off = beta_offset;
  d.tc_off = tc_offset;
  d.sub = subblock_size;
  d.y_begin = y_begin;
  d.y_end = y_end;
  d.cus = d_cus;
  d.map = d_cu_map;
  d.map_stride = map_stride;
  d.map_rows = (rec->h + 3) / 4;
  d.comp_mask = comp_mask;
  const int nx = (rec->w + subblock_size - 1) / subblock_size;
  const int ny = (y_end - y_begin + subblock_size - 1) / subblock_size;
  const dim3 grid((nx + 63) / 64, ny);
  if (pass == 0)
    hipLaunchKernelGGL(deblock_pass_kernel<true>, grid, dim3(64), 0, ctx->stream, d,
                       rec->v);
  else
    hipLaunchKernelGGL(deblock_pass_kernel<false>, grid, dim3(64), 0, ctx->stream, d,
                       rec->v);
  CHECK_LAUNCH(ctx, "deblock_rows");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_deblock_rows(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                  const xvcgpu_cu_info *d_cus, int n_cus,
                                  const int32_t *d_cu_map, int map_stride,
                                  int pic_is_bipred, int beta_offset,
                                  int tc_offset, int subblock_size, int pass,
                                  int y_begin, int y_end) {
  return deblock_rows_masked(ctx, rec, d_cus, n_cus, d_cu_map, map_stride, pic_is_bipred,
                             beta_offset, tc_offset, subblock_size, pass, y_begin, y_end, 3);
}

xvcgpu_status xvcgpu_deblock_tree(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                  const xvcgpu_cu_info *d_cus, int n_cus,
                                  const int32_t *d_cu_map, int map_stride,
                                  int pic_is_bipred, int beta_offset, int tc_offset,
                                  int subblock_size, int comp_mask) {
  if (!rec || comp_mask < 1 || comp_mask > 3) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_status st = deblock_rows_masked(ctx, rec, d_cus, n_cus, d_cu_map, map_stride,
                                         pic_is_bipred, beta_offset, tc_offset,
                                         subblock_size, 0, 0, rec->h, comp_mask);
  if (st != XVCGPU_OK) return st;
  return deblock_rows_masked(ctx, rec, d_cus, n_cus, d_cu_map, map_stride, pic_is_bipred,
                             beta_offset, tc_offset, subblock_size, 1, 0, rec->h, comp_mask);
}

xvcgpu_status xvcgpu_deblock(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                             const xvcgpu_cu_info *d_cus, int n_cus,
                             const int32_t *d_cu_map, int map_stride,
                             int pic_is_bipred, int beta_offset, int tc_offset,
                             int subblock_size) {
  if (!rec) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_status st = xvcgpu_deblock_rows(ctx, rec, d_cus, n_cus, d_cu_map, map_stride,
                                         pic_is_bipred, beta_offset, tc_offset,
                                         subblock_size, 0, 0, rec->h);
  if (st != XVCGPU_OK) return st;
  return xvcgpu_deblock_rows(ctx, rec, d_cus, n_cus, d_cu_map, map_stride,
                             pic_is_bipred, beta_offset, tc_offset, subblock_size, 1,
                             0, rec->h);
}

// Scratch for the per-block results of xvcgpu_picture_ssd; grown when a larger
// picture is created, so no allocation happens on the measurement path.
static xvcgpu_status ensure_ssd_part(xvcgpu_ctx *ctx, int items) {
  if (items <= ctx->ssd_part_cap) return XVCGPU_OK;
  if (ctx->d_ssd_part) {
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_ssd_part);
    ctx->d_ssd_part = nullptr;
    ctx->ssd_part_cap = 0;
  }
  hipError_t e = hipMalloc(&ctx->d_ssd_part, sizeof(unsigned long long) * 2 * items);
  if (e != hipSuccess) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "hipMalloc", e);
  ctx->ssd_part_cap = items;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_ssd(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                 const xvcgpu_picture *b, int comp,
                                 int shift_bitdepth, uint64_t *d_out) {
  return xvcgpu_picture_ssd_rows(ctx, a, b, comp, shift_bitdepth, 0, 1 << 30, d_out);
}

xvcgpu_status xvcgpu_picture_ssd_rows(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                      const xvcgpu_picture *b, int comp,
                                      int shift_bitdepth, int y_begin, int y_end,
                                      uint64_t *d_out) {
  if (!ctx || !a || !b || comp < 0 || comp > 2 || !d_out || shift_bitdepth < 8)
    return XVCGPU_INVALID_ARGUMENT;
  if (a->w != b->w || a->h != b->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  const PlaneView pa = a->v.c[comp], pb = b->v.c[comp];
  const int items = ssd_items(pa.w, pa.h);
  {
    const xvcgpu_status st = ensure_ssd_part(ctx, items);
    if (st != XVCGPU_OK) return st;
  }
  if (items > 0)
    hipLaunchKernelGGL(picture_ssd_kernel, dim3(items), dim3(256), 0, ctx->stream, pa,
                       pb, 2 * (shift_bitdepth - 8), y_begin, y_end, ctx->d_ssd_part);
  hipLaunchKernelGGL(picture_ssd_sum_kernel, dim3(1), dim3(256), 0, ctx->stream,
                     ctx->d_ssd_part, items,
                     reinterpret_cast<unsigned long long *>(d_out));
  CHECK_LAUNCH(ctx, "picture_ssd");
  return XVCGPU_OK;
}

// Scratch of xvcgpu_deblock_pad_ssd: per-tile results; sized when a picture is
// created.
static xvcgpu_status ensure_tail(xvcgpu_ctx *ctx, int tiles) {
  if (tiles <= ctx->tail_cap) return XVCGPU_OK;
  if (ctx->d_tail_part) {
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_tail_part);
    ctx->d_tail_part = nullptr;
    ctx->tail_cap = 0;
  }
  const size_t bytes = sizeof(unsigned long long) * (2 * (size_t)tiles + 2);
  hipError_t e = hipMalloc(&ctx->d_tail_part, bytes);
  if (e != hipSuccess) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "hipMalloc", e);
  hipMemsetAsync(ctx->d_tail_part, 0, bytes, ctx->stream);
  ctx->tail_cap = tiles;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_deblock_pad_ssd(xvcgpu_ctx *ctx, const xvcgpu_picture *src,
                                     xvcgpu_picture *dst, const xvcgpu_picture *orig,
                                     const xvcgpu_cu_info *d_cus, int n_cus,
                                     const int32_t *d_cu_map, int map_stride,
                                     int pic_is_bipred, int beta_offset, int tc_offset,
                                     int shift_bitdepth, uint64_t *d_ssd) {
  if (!ctx || !src || !dst || src == dst || src->base == dst->base || !d_cus || n_cus <= 0 ||
      !d_cu_map || map_stride < (dst->w + 3) / 4 || (orig && (!d_ssd || shift_bitdepth < 8)))
    return XVCGPU_INVALID_ARGUMENT;
  if (src->w != dst->w || src->h != dst->h || src->bd != dst->bd ||
      (orig && (orig->w != dst->w || orig->h != dst->h)))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if ((dst->w & 7) || (dst->h & 7))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "deblock_pad_ssd: picture size not a multiple of 8");
  const int tiles = tail_tiles(dst->w, dst->h);
  {
    const xvcgpu_status st = ensure_tail(ctx, tiles);
    if (st != XVCGPU_OK) return st;
  }
  DbParams d;
  d.bd = dst->bd;
  d.pic_w = dst->w;
  d.pic_h = dst->h;
  d.bipred = pic_is_bipred;
  d.beta_off = beta_offset;
  d.tc_off = tc_offset;
  d.sub = 4;
  d.y_begin = 0;
  d.y_end = dst->h;
  d.cus = d_cus;
  d.map = d_cu_map;
  d.map_stride = map_stride;
  d.map_rows = (dst->h + 3) / 4;
  d.comp_mask = 3;
  unsigned long long *part = ctx->d_tail_part;
  if (orig) {
    hipLaunchKernelGGL(deblock_tail_kernel<true>, dim3((tiles + 7) / 8 * 8), dim3(256), 0, ctx->stream, d,
                       src->v, dst->v, orig->v.c[0], 2 * (shift_bitdepth - 8), part);
    hipLaunchKernelGGL(picture_ssd_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, part, tiles,
                       reinterpret_cast<unsigned long long *>(d_ssd));
  } else {
    hipLaunchKernelGGL(deblock_tail_kernel<false>, dim3((tiles + 7) / 8 * 8), dim3(256), 0, ctx->stream, d,
                       src->v, dst->v, dst->v.c[0], 0, part);
  }
  CHECK_LAUNCH(ctx, "deblock_pad_ssd");
  return XVCGPU_OK;
}

/* ---- whole-picture passes around the hot path (k_stats.h) ---- */
static const int kStatsHistWords = 4096;

// Scratch of the statistics passes; like the SSD scratch it is sized when a
// picture is created, never on a measurement path.
static xvcgpu_status ensure_stats(xvcgpu_ctx *ctx, int rows) {
  if (rows <= ctx->stats_rows_cap) return XVCGPU_OK;
  if (ctx->d_stats) {
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_stats);
    ctx->d_stats = nullptr;
    ctx->stats_rows_cap = 0;
  }
  const size_t bytes = sizeof(uint32_t) * ((size_t)kStatsHistWords + rows);
  hipError_t e = hipMalloc(&ctx->d_stats, bytes);
  if (e != hipSuccess) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "hipMalloc", e);
  hipMemsetAsync(ctx->d_stats, 0, bytes, ctx->stream);
  ctx->stats_rows_cap = rows;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_import(xvcgpu_ctx *ctx, xvcgpu_picture *pic,
                                    const void *d_src, int in_width, int in_height,
                                    int in_bitdepth) {
  if (!ctx || !pic || !d_src) return XVCGPU_INVALID_ARGUMENT;
  if (in_width < 2 || in_height < 2 || (in_width & 1) || (in_height & 1) ||
      in_width > pic->w || in_height > pic->h || in_bitdepth < 8 ||
      in_bitdepth > pic->bd || in_bitdepth > 16)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "import: input size/bitdepth");
  ImportArgs a;
  const size_t bps = in_bitdepth > 8 ? 2 : 1;
  const uint8_t *src = static_cast<const uint8_t *>(d_src);
  for (int c = 0; c < 3; c++) {
    a.src[c] = src;
    a.in_w[c] = c ? in_width >> 1 : in_width;
    a.in_h[c] = c ? in_height >> 1 : in_height;
    src += (size_t)a.in_w[c] * a.in_h[c] * bps;
  }
  a.wide = in_bitdepth > 8;
  a.upshift = pic->bd - in_bitdepth;
  hipLaunchKernelGGL(picture_import_kernel, dim3(pic->h, 3), dim3(256), 0, ctx->stream,
                     pic->v, a);
  CHECK_LAUNCH(ctx, "picture_import");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_export(xvcgpu_ctx *ctx, const xvcgpu_picture *pic,
                                    void *d_dst, int display_width, int display_height,
                                    int out_bitdepth, int dither) {
  if (!ctx || !pic || !d_dst) return XVCGPU_INVALID_ARGUMENT;
  {  // the scratch is sized by the pictures created on THIS context; a picture of
     // another context (or a larger one) must not run into it
    const xvcgpu_status st_ = ensure_stats(ctx, 2 * pic->h);
    if (st_ != XVCGPU_OK) return st_;
  }
  if (display_width < 2 || display_height < 2 || (display_width & 1) ||
      (display_height & 1) || display_width > pic->w || display_height > pic->h ||
      out_bitdepth < 1 || out_bitdepth > 16)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "export: output size/bitdepth");
  ExportArgs a;
  a.wide = out_bitdepth > 8;
  const size_t bps = a.wide ? 2 : 1;
  uint8_t *dst = static_cast<uint8_t *>(d_dst);
  int rows = 0;
  for (int c = 0; c < 3; c++) {
    a.dst[c] = dst;
    a.w[c] = c ? display_width >> 1 : display_width;
    a.h[c] = c ? display_height >> 1 : display_height;
    a.row_base[c] = rows;
    rows += a.h[c];
    dst += (size_t)a.w[c] * a.h[c] * bps;
  }
  // CopyToBytesWithShift's dispatch (resample.cc:304-338)
  a.smax = (1 << out_bitdepth) - 1;
  if (out_bitdepth >= pic->bd || (!a.wide && pic->bd <= 8)) {
    a.mode = 0;
    a.shift = a.wide ? out_bitdepth - pic->bd : 0;
  } else {
    a.mode = dither ? 2 : 1;
    a.shift = pic->bd - out_bitdepth;
  }
  uint32_t *row_words = ctx->d_stats + kStatsHistWords;
  a.row_carry = row_words;
  if (a.mode == 2) {
    hipLaunchKernelGGL(export_row_sums_kernel, dim3(a.h[0], 3), dim3(256), 0, ctx->stream,
                       pic->v, a, row_words);
    hipLaunchKernelGGL(export_row_scan_kernel, dim3(3), dim3(256), 0, ctx->stream, a,
                       row_words);
  }
  hipLaunchKernelGGL(picture_export_kernel, dim3(a.h[0], 3), dim3(256), 0, ctx->stream,
                     pic->v, a);
  CHECK_LAUNCH(ctx, "picture_export");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_picture_crc(xvcgpu_ctx *ctx, const xvcgpu_picture *pic, int mode,
                                 uint8_t *d_hash) {
  if (!ctx || !pic || !d_hash || mode < 0 || mode > 1) return XVCGPU_INVALID_ARGUMENT;
  {  // the scratch is sized by the pictures created on THIS context; a picture of
     // another context (or a larger one) must not run into it
    const xvcgpu_status st_ = ensure_stats(ctx, 2 * pic->h);
    if (st_ != XVCGPU_OK) return st_;
  }
  static const CrcPow2 pow2 = [] {
    CrcPow2 t;
    uint32_t v = 2;  // x
    for (int i = 0; i < 48; i++) {
      t.v[i] = (uint16_t)v;
      v = crc_mulmod(v, v);
    }
    return t;
  }();
  CrcArgs a;
  a.pow2 = pow2;
  a.wide = pic->bd > 8;
  a.mode = mode;
  if (!ctx->d_crc_tables) {
    static CrcTables host[2];
    crc_build_tables(host[0], pow2, 0);
    crc_build_tables(host[1], pow2, 1);
    hipError_t e = hipMalloc(&ctx->d_crc_tables, sizeof(host));
    if (e != hipSuccess) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "hipMalloc", e);
    HIP_TRY(ctx, hipMemcpy(ctx->d_crc_tables, host, sizeof(host), hipMemcpyHostToDevice));
  }
  // one word per workgroup and plane in the row scratch (3 * ceil(h / 4) <= 2 * h)
  uint32_t *words = ctx->d_stats + kStatsHistWords;
  const int n_wg = (pic->h + 3) / 4;
  hipLaunchKernelGGL(crc_rows_kernel, dim3(n_wg, 3), dim3(256), 0, ctx->stream, pic->v, a,
                     ctx->d_crc_tables + a.wide, words);
  hipLaunchKernelGGL(crc_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, pic->v, mode,
                     n_wg, words, d_hash);
  CHECK_LAUNCH(ctx, "picture_crc");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_variance_map(xvcgpu_ctx *ctx, const xvcgpu_picture *pic,
                                  uint64_t *d_var16, int ctu_size, uint64_t *d_ctu_var) {
  if (!ctx || !pic || !d_var16) return XVCGPU_INVALID_ARGUMENT;
  if (d_ctu_var && ctu_size != 16 && ctu_size != 32 && ctu_size != 64 && ctu_size != 128)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "variance_map: ctu size");
  const int bw = (pic->w + 15) / 16, bh = (pic->h + 15) / 16;
  hipLaunchKernelGGL(variance_map_kernel, dim3((bw * bh + 3) / 4), dim3(256), 0,
                     ctx->stream, pic->v.c[0], bw, bw * bh,
                     reinterpret_cast<unsigned long long *>(d_var16));
  if (d_ctu_var) {
    const int n = ((pic->w + ctu_size - 1) / ctu_size) * ((pic->h + ctu_size - 1) / ctu_size);
    hipLaunchKernelGGL(ctu_variance_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long *>(d_var16), pic->w, pic->h,
                       ctu_size, reinterpret_cast<unsigned long long *>(d_ctu_var));
  }
  CHECK_LAUNCH(ctx, "variance_map");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_histogram_distance(xvcgpu_ctx *ctx, const xvcgpu_picture *a,
                                        const xvcgpu_picture *b, int64_t *d_out) {
  if (!ctx || !a || !b || !d_out) return XVCGPU_INVALID_ARGUMENT;
  {  // the scratch is sized by the pictures created on THIS context; a picture of
     // another context (or a larger one) must not run into it
    const xvcgpu_status st_ = ensure_stats(ctx, 2 * a->h);
    if (st_ != XVCGPU_OK) return st_;
  }
  if (a->w != b->w || a->h != b->h || a->bd != b->bd || a->bd > 12)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  const int buckets = 1 << a->bd;
  const int rows_per_wg = (a->h + 511) / 512;
  const int n_wg = (a->h + rows_per_wg - 1) / rows_per_wg;
  int *hist = reinterpret_cast<int *>(ctx->d_stats);
  hipLaunchKernelGGL(histogram_diff_kernel, dim3(n_wg), dim3(256), sizeof(int) * buckets,
                     ctx->stream, a->v.c[0], b->v.c[0], buckets, rows_per_wg, hist);
  hipLaunchKernelGGL(histogram_abs_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, hist,
                     buckets, reinterpret_cast<long long *>(d_out));
  CHECK_LAUNCH(ctx, "histogram_distance");
  return XVCGPU_OK;
}


/* ---- intra prediction / SATD mode pre-selection (k_intra.h) ---- */
xvcgpu_status xvcgpu_intra_pred_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *rec,
                                      xvcgpu_picture *pred,
                                      const xvcgpu_intra_block *d_jobs, int n) {
  if (!ctx || !rec || !pred || (!d_jobs && n > 0) || n < 0) return XVCGPU_INVALID_ARGUMENT;
  if (rec->w != pred->w || rec->h != pred->h || rec->bd != pred->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(intra_pred_kernel, dim3(n), dim3(256), 0, ctx->stream, rec->v, pred->v,
                     d_jobs, n);
  CHECK_LAUNCH(ctx, "intra_pred_batch");
  return XVCGPU_OK;
}

#ifndef XVCGPU_INTRA_WAVES_GRID
#define XVCGPU_INTRA_WAVES_GRID 128
#endif
xvcgpu_status xvcgpu_intra_recon_waves(xvcgpu_ctx *ctx, xvcgpu_picture *rec,
                                       xvcgpu_picture *pred,
                                       const xvcgpu_intra_block *d_jobs,
                                       const xvcgpu_tx_block *d_blocks,
                                       const int32_t *d_wave_first, int n_waves,
                                       const int16_t *d_levels,
                                       const uint32_t *d_level_offsets,
                                       const int32_t *d_nnz) {
  if (!ctx || !rec || !pred || n_waves < 0 ||
      (n_waves && (!d_jobs || !d_blocks || !d_wave_first || !d_levels || !d_level_offsets ||
                   !d_nnz)))
    return XVCGPU_INVALID_ARGUMENT;
  if (rec->w != pred->w || rec->h != pred->h || rec->bd != pred->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n_waves == 0) return XVCGPU_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // a wave of a 1080p picture holds a few dozen jobs: a fraction of the chip's
  // workgroup slots is enough (all of them must be resident together)
  int &grid = ctx->intra_waves_grid;   // 0: not asked yet, < 0: refused
  if (grid == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, intra_waves_kernel, 256, 0) !=
            hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) !=
            hipSuccess ||
        per_cu < 1 || cus < 1)
      grid = -1;
    else
      grid = cus * per_cu < XVCGPU_INTRA_WAVES_GRID ? cus * per_cu : XVCGPU_INTRA_WAVES_GRID;
  }
  if (grid < 0) return XVCGPU_UNSUPPORTED;
  // the waves' counters
  if (ctx->intra_done_cap < n_waves) {
    if (ctx->d_intra_done) hipFree(ctx->d_intra_done);
    ctx->d_intra_done = nullptr;
    ctx->intra_done_cap = 0;
    if (hipMalloc(&ctx->d_intra_done, sizeof(int) * (size_t)(n_waves + 256)) != hipSuccess)
      return XVCGPU_OUT_OF_MEMORY;
    ctx->intra_done_cap = n_waves + 256;
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_intra_done, 0, sizeof(int) * (size_t)n_waves, ctx->stream));
  int *done = ctx->d_intra_done;
  PicView rv = rec->v, pv = pred->v;
  int16_t *lv = const_cast<int16_t *>(d_levels);
  int32_t *nz = const_cast<int32_t *>(d_nnz);
  const int16_t *tables = ctx->d_tx_tables;
  TxTableLayout lay = xvcgpu_tx_layout();
  void *args[] = {&rv, &pv, &d_jobs, &d_blocks, &d_wave_first, &n_waves, &lv, &d_level_offsets,
                  &nz, &tables, &lay, &done};
  if (hipLaunchCooperativeKernel(reinterpret_cast<const void *>(intra_waves_kernel), dim3(grid),
                                 dim3(256), args, 0, ctx->stream) != hipSuccess) {
    (void)hipGetLastError();
    return XVCGPU_UNSUPPORTED;
  }
  CHECK_LAUNCH(ctx, "intra_recon_waves");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_intra_satd_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                      const xvcgpu_picture *rec,
                                      const xvcgpu_intra_block *d_jobs, int n,
                                      uint32_t *d_dist, int max_block_size) {
  if (!ctx || !orig || !rec || (!d_jobs && n > 0) || n < 0 || (!d_dist && n > 0) ||
      max_block_size < 4 || max_block_size > 64)
    return XVCGPU_INVALID_ARGUMENT;
  if (rec->w != orig->w || rec->h != orig->h || rec->bd != orig->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  // workgroups per job: enough to put ~2 workgroups on every CU, at most one
  // mode per wave (67 modes / 4 waves -> 17)
  int split = (2 * 256 + n - 1) / n;
  split = split < 1 ? 1 : (split > 17 ? 17 : split);
  const dim3 grid(n, split);
  if (max_block_size <= 16)
    hipLaunchKernelGGL(intra_satd_kernel<16>, grid, dim3(256), 0, ctx->stream, orig->v,
                       rec->v, d_jobs, n, d_dist);
  else if (max_block_size <= 32)
    hipLaunchKernelGGL(intra_satd_kernel<32>, grid, dim3(256), 0, ctx->stream, orig->v,
                       rec->v, d_jobs, n, d_dist);
  else
    hipLaunchKernelGGL(intra_satd_kernel<64>, grid, dim3(256), 0, ctx->stream, orig->v,
                       rec->v, d_jobs, n, d_dist);
  CHECK_LAUNCH(ctx, "intra_satd_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_intra_recon_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                       xvcgpu_picture *rec,
                                       const xvcgpu_intra_block *d_jobs,
                                       const xvcgpu_tx_block *d_blocks, int n,
                                       int16_t *d_levels, const uint32_t *d_level_offsets,
                                       int32_t *d_nnz) {
  if (!ctx || !rec || n < 0 || (n && (!d_jobs || !d_blocks))) return XVCGPU_INVALID_ARGUMENT;
  // decoder form (no original): the levels are an input and must be there
  if (!orig && n && (!d_levels || !d_level_offsets || !d_nnz)) return XVCGPU_INVALID_ARGUMENT;
  if (orig && (orig->w != rec->w || orig->h != rec->h || orig->bd != rec->bd))
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  const dim3 grid((n + TX2_WAVES - 1) / TX2_WAVES), block(64 * TX2_WAVES);
  if (orig)
    hipLaunchKernelGGL(intra_recon_wave_kernel<TX_MODE_FULL>, grid, block, 0, ctx->stream,
                       orig->v, rec->v, d_jobs, d_blocks, n, d_levels, d_level_offsets, d_nnz,
                       ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout());
  else
    hipLaunchKernelGGL(intra_recon_wave_kernel<TX_MODE_INV>, grid, block, 0, ctx->stream,
                       rec->v, rec->v, d_jobs, d_blocks, n, d_levels, d_level_offsets, d_nnz,
                       ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout());
  CHECK_LAUNCH(ctx, "intra_recon_batch");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_intra_select_modes(xvcgpu_ctx *ctx, const uint32_t *d_dist,
                                        const uint32_t *d_mode_cost, int n,
                                        int32_t *d_modes, xvcgpu_intra_block *d_jobs,
                                        xvcgpu_tx_block *d_blocks, int per_cu) {
  if (!ctx || n < 0 || (n && !d_dist) || per_cu < 0 || per_cu > 3 ||
      ((d_jobs || d_blocks) && per_cu == 0))
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(intra_select_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream,
                     d_dist, d_mode_cost, n, d_modes, d_jobs, d_blocks, per_cu);
  CHECK_LAUNCH(ctx, "intra_select_modes");
  return XVCGPU_OK;
}

/* ---- a picture per call ---- */
xvcgpu_status xvcgpu_frame_pass(xvcgpu_ctx *ctx, const xvcgpu_frame_pass_args *a,
                                int phases) {
  if (!ctx || !a || !a->rec) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_status st = XVCGPU_OK;
  // the five launches of the tail as one (k_tail.h)
  const int kAll = XVC_FP_ENCODE | XVC_FP_DEBLOCK_V | XVC_FP_DEBLOCK_H | XVC_FP_PAD | XVC_FP_SSD;
  const bool fused_tail = a->scratch_rec && (phases & kAll) == kAll && a->n_cus > 0 &&
                          a->n_cus == a->n_cus_total &&
                          a->db_y_begin == 0 && a->db_y_end >= a->rec->h &&
                          a->dbh_y_end >= a->rec->h && a->ssd_y_begin == 0 &&
                          a->ssd_y_end >= a->rec->h && !(a->rec->w & 7) && !(a->rec->h & 7);
  xvcgpu_picture *const rec = fused_tail ? a->scratch_rec : a->rec;
  hipStream_t main_stream = nullptr;   // set while the pass runs on ctx->hi_stream
  struct Back {
    xvcgpu_ctx *c;
    hipStream_t *m;
    ~Back() {   // every way out: the chain continues on its own stream, after the tail
      if (!*m) return;
      hipEventRecord(c->ev_hi_out, c->hi_stream);
      c->stream = *m;
      hipStreamWaitEvent(c->stream, c->ev_hi_out, 0);
    }
  } back = {ctx, &main_stream};
  if ((phases & XVC_FP_ENCODE) && a->n_cus > 0) {
    // the pass's jobs are the CUs of its grid: where the caller vouches that they are all
    // 16x16 (16x8 in the bottom row of a 1080-line picture) the search's exact-shape kernel
    // (a pass of smaller CUs must not take it: its jobs would all be left to the few waves
    // of the leftover kernel)
    st = xvcgpu_me_search_sized(ctx, a->orig, a->ref,
                                XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL |
                                    (a->me_only_sq16 ? XVCGPU_ME_HINT_SQ16 | XVCGPU_ME_ONLY_SQ16 : 0),
                                a->d_me, a->n_cus, a->d_results, a->max_block_size);
    if (st != XVCGPU_OK) return st;
    if (a->d_rdoq_params && a->pred) {
      bool in_place = false, classified = false;
      if (a->max_block_size <= 16 && a->n_tx == 3 * a->n_cus) {
        // prediction + forward transform in one kernel (transform blocks in CU
        // order, Y U V each: block 3 * cu + comp)
        // (the prediction goes straight into the reconstruction's picture: the
        // inverse half then works in place and skips the blocks without levels)
        // and it classifies the blocks for the quantiser on the way (the
        // coefficients are at hand: no separate pass over all of them)
        // ... and proves the blocks it can all zero (k_rdoq.h)
        st = xvcgpu_fwd_from_me_classify_prove(ctx, a->orig, a->ref, rec, a->d_me, a->d_results,
                                               a->n_cus, a->qp_y, a->qp_c, a->ref_poc,
                                               a->d_coeffs, a->d_level_off, a->n_coeffs,
                                               a->d_levels, a->d_nnz, a->d_cus_own,
                                               a->d_rdoq_contexts, a->d_rdoq_params);
        in_place = classified = true;
      } else {
        st = xvcgpu_mc_from_me(ctx, a->ref, a->pred, a->d_me, a->d_results, a->n_cus);
        if (st == XVCGPU_OK)
          st = xvcgpu_fwd_transform_batch(ctx, a->orig, a->pred, a->d_tx, a->n_tx, a->d_coeffs,
                                          a->d_level_off);
      }
      // the caller's word about its blocks holds for this call's batch
      const int four_before = ctx->rdoq_four_lane_only;
      if (st == XVCGPU_OK && a->tx_four_lane_only && !four_before)
        st = xvcgpu_quant_rdo_set_four_lane_only(ctx, 1);
      if (st == XVCGPU_OK)
        st = classified
                 ? xvcgpu_quant_rdo_classified_batch(ctx, a->rec->bd, a->d_tx, a->n_tx,
                                                     a->d_coeffs, a->d_level_off, a->n_coeffs,
                                                     a->d_levels, a->d_nnz, a->d_rdoq_contexts,
                                                     a->d_rdoq_params, a->d_cus_own)
                 : xvcgpu_quant_rdo_batch(ctx, a->rec->bd, a->d_tx, a->n_tx, a->d_coeffs,
                                          a->d_level_off, a->n_coeffs, a->d_levels, a->d_nnz,
                                          a->d_rdoq_contexts, a->d_rdoq_params);
      ctx->rdoq_four_lane_only = four_before;
      if (st == XVCGPU_OK) {
        if (!a->d_tx || !a->d_levels || !a->d_level_off || !a->d_nnz) {
          st = XVCGPU_INVALID_ARGUMENT;
        } else if (a->n_tx > 0) {
          const PicView &pv = in_place ? rec->v : a->pred->v;
          if (ctx->hi_stream && fused_tail && in_place) {
            // the rest of the pass - inverse transform and the fused tail, two short
            // kernels - on the high-priority stream: beside other pictures' searches
            // (long-lived waves on every CU) their workgroups otherwise wait for slots
            // many times their own duration (profiles/r03_bench_4320p_kernel_stats.csv:
            // the tail 2152 us in flight against 118 alone)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_hi_in, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->hi_stream, ctx->ev_hi_in, 0));
            main_stream = ctx->stream;
            ctx->stream = ctx->hi_stream;
          }
          if (in_place) {
            // blocks 3 * cu + comp, CUs up to 16x16 (the condition of in_place): the
            // U and V blocks of a CU share a wave
            st = xvcgpu_inv_transform_cu_order(ctx, rec, a->d_tx, a->n_cus, a->d_levels,
                                               a->d_level_off, a->d_nnz);
          } else {
            launch_residual<TX_MODE_INV>(ctx, pv, pv, rec->v, a->d_tx, a->n_tx, a->d_levels,
                                         a->d_level_off, a->d_nnz, nullptr, false);
          }
          CHECK_LAUNCH(ctx, "inv_transform_batch");
        }
      }
      if (st == XVCGPU_OK && !(classified && a->d_cus_own))   // (else written on the way)
        st = xvcgpu_cu_info_from_me(ctx, a->d_me, a->d_results, a->d_nnz, a->d_luma_tx_index,
                                    a->n_cus, a->qp_y, a->qp_c, a->ref_poc, a->d_cus_own);
    } else if (a->d_rdoq_params)
      st = xvcgpu_recon_from_me_rdoq(ctx, a->orig, a->ref, rec, a->d_me, a->d_results,
                                     a->n_cus, a->qp_y, a->qp_c, 0, a->ref_poc, a->d_nnz,
                                     a->d_cus_own, a->d_rdoq_contexts, a->d_rdoq_params);
    else
      st = xvcgpu_recon_from_me(ctx, a->orig, a->ref, rec, a->d_me, a->d_results, a->n_cus,
                                a->qp_y, a->qp_c, 0, a->ref_poc, a->d_nnz, a->d_cus_own);
    if (st != XVCGPU_OK) return st;
  }
  if (fused_tail)
    return xvcgpu_deblock_pad_ssd(ctx, a->scratch_rec, a->rec, a->orig, a->d_cus, a->n_cus_total,
                                  a->d_cu_map, a->map_stride, 0, 0, 0, a->shift_bitdepth,
                                  a->d_ssd);
  if (phases & XVC_FP_DEBLOCK_V) {
    st = xvcgpu_deblock_rows(ctx, a->rec, a->d_cus, a->n_cus_total, a->d_cu_map, a->map_stride,
                             0, 0, 0, 4, 0, a->db_y_begin, a->db_y_end);
    if (st != XVCGPU_OK) return st;
  }
  if (phases & XVC_FP_DEBLOCK_H) {
    st = xvcgpu_deblock_rows(ctx, a->rec, a->d_cus, a->n_cus_total, a->d_cu_map, a->map_stride,
                             0, 0, 0, 4, 1, a->db_y_begin, a->dbh_y_end);
    if (st != XVCGPU_OK) return st;
  }
  if (phases & XVC_FP_PAD) {
    st = xvcgpu_pad_border(ctx, a->rec);
    if (st != XVCGPU_OK) return st;
  }
  if (phases & XVC_FP_SSD)
    st = xvcgpu_picture_ssd_rows(ctx, a->orig, a->rec, 0, a->shift_bitdepth, a->ssd_y_begin,
                                 a->ssd_y_end, a->d_ssd);
  return st;
}

/* ---- several pictures per call: every kernel launched once for all of them ---- */
xvcgpu_status xvcgpu_frame_pass_multi(xvcgpu_ctx *const *ctxs,
                                      const xvcgpu_frame_pass_args *const *args, int n,
                                      int phases) {
  if (!ctxs || !args || n < 1 || !ctxs[0]) return XVCGPU_INVALID_ARGUMENT;
  xvcgpu_ctx *ctx = ctxs[0];
  const int kAll = XVC_FP_ENCODE | XVC_FP_DEBLOCK_V | XVC_FP_DEBLOCK_H | XVC_FP_PAD | XVC_FP_SSD;
  // the form the launches below cover: whole pictures of CUs up to 16x16 (and at
  // least 8x8: scratch_rec), all phases, the packed RDOQ pipeline or QuantFast;
  // anything else runs picture by picture, each on its own context
  bool batched = n >= 2 && n <= XVC_MULTI_MAX && (phases & kAll) == kAll;
  for (int i = 0; i < n && batched; i++) {
    const xvcgpu_frame_pass_args *a = args[i];
    if (!ctxs[i] || !a || !a->orig || !a->ref || !a->rec) return XVCGPU_INVALID_ARGUMENT;
    const bool rdoq_packed = a->d_rdoq_params && a->pred && a->n_tx == 3 * a->n_cus;
    const bool fast = !a->d_rdoq_params;
    batched = a->scratch_rec && a->n_cus > 0 && a->n_cus == a->n_cus_total &&
              a->max_block_size <= 16 && (rdoq_packed || fast) &&
              (rdoq_packed == (args[0]->d_rdoq_params != nullptr)) && a->db_y_begin == 0 &&
              a->db_y_end >= a->rec->h && a->dbh_y_end >= a->rec->h && a->ssd_y_begin == 0 &&
              a->ssd_y_end >= a->rec->h && !(a->rec->w & 7) && !(a->rec->h & 7) &&
              a->rec->w == args[0]->rec->w && a->rec->h == args[0]->rec->h &&
              a->rec->bd == args[0]->rec->bd && ctxs[i]->device == ctx->device &&
              // what the launches below dereference (the single-picture path
              // validates the same pointers, so a picture missing one goes there)
              a->d_me && a->d_results && a->d_cus && a->d_cu_map && a->d_ssd && a->d_nnz &&
              a->d_cus_own &&
              (!rdoq_packed || (a->d_tx && a->d_levels && a->d_coeffs && a->d_level_off &&
                                a->d_luma_tx_index && a->d_rdoq_contexts));
  }
  if (!batched) {
    // Picture by picture.  The batched form runs everything on ctxs[0]'s
    // stream; so that a caller sees ONE ordering rule whichever form a call
    // takes, a picture whose context has another stream is fenced into that
    // stream: it starts after what ctxs[0]'s stream holds now, and ctxs[0]'s
    // stream continues only after it.
    for (int i = 0; i < n; i++) {
      if (!ctxs[i] || !args[i]) return XVCGPU_INVALID_ARGUMENT;
      const bool foreign = ctxs[i]->stream != ctx->stream;
      hipEvent_t before = nullptr, after = nullptr;
      // an event belongs to the device that was current when it was created and
      // can only be recorded on that device's streams: `before` is ctxs[0]'s,
      // `after` the picture's own (contexts on another device are exactly what
      // sends a call down this path); both are destroyed on every way out
      auto fence_in = [&]() -> hipError_t {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&before, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(before, ctx->stream);
        if (e == hipSuccess) e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&after, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctxs[i]->stream, before, 0);
        return e;
      };
      auto fence_out = [&]() -> hipError_t {
        hipError_t e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipEventRecord(after, ctxs[i]->stream);
        if (e == hipSuccess) e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, after, 0);
        return e;
      };
      hipError_t herr = foreign ? fence_in() : hipSuccess;
      xvcgpu_status st = XVCGPU_OK;
      if (herr == hipSuccess) {
        st = xvcgpu_frame_pass(ctxs[i], args[i], phases);
        if (foreign && st == XVCGPU_OK) herr = fence_out();
      }
      if (before) hipEventDestroy(before);   // released when the recorded work has passed them
      if (after) hipEventDestroy(after);
      HIP_TRY(ctx, herr);
      if (st != XVCGPU_OK) return st;
    }
    return XVCGPU_OK;
  }
  const bool rdoq = args[0]->d_rdoq_params != nullptr;
  const dim3 one(1);
  int max_cus = 0, max_tx = 0;
  for (int i = 0; i < n; i++) {
    max_cus = std::max(max_cus, (int)args[i]->n_cus);
    max_tx = std::max(max_tx, (int)args[i]->n_tx);
  }
  const TxTableLayout lay = xvcgpu_tx_layout();
  // 1. the motion searches
  {
    MultiArgs<MeMultiArgs> m;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      xvcgpu_ctx *c = ctxs[i];
      const int e = c->me_epoch = (c->me_epoch + 1) % 3;   // the picture's own rotation records
      MeMultiArgs &k = m.a[i];
      k.orig = a->orig->v;
      k.ref = a->ref->v;
      k.blocks = a->d_me;
      k.n = a->n_cus;
      k.results = a->d_results;
      k.sched.use = c->d_me_rot + e % 3;
      k.sched.record = c->d_me_rot + (e + 1) % 3;
      k.sched.clear = c->d_me_rot + (e + 2) % 3;
    }
    hipLaunchKernelGGL(me_search_multi_kernel, dim3(me2_grid(max_cus, ME2_WAVES(16)).x, n),
                       dim3(64 * ME2_WAVES(16)), 0, ctx->stream, m, ctx->d_tz_pattern);
  }
  // 2. prediction + transform (RDOQ: forward half only, the quantiser follows)
  {
    MultiArgs<ReconMultiArgs> m;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      ReconMultiArgs &k = m.a[i];
      k.orig = a->orig->v;
      k.ref = a->ref->v;
      k.rec = a->scratch_rec->v;   // (RDOQ: the prediction; the inverse half works in place)
      k.blocks = a->d_me;
      k.results = a->d_results;
      k.n_cus = a->n_cus;
      k.qp_y = rdoq ? 0 : a->qp_y;
      k.qp_c = rdoq ? 0 : a->qp_c;
      k.ref_poc = rdoq ? 0 : a->ref_poc;
      k.nnz_out = rdoq ? nullptr : a->d_nnz;
      k.cus = rdoq ? nullptr : a->d_cus_own;
      k.coeffs = rdoq ? a->d_coeffs : nullptr;
      k.coeff_off = rdoq ? a->d_level_off : nullptr;
    }
    const int n_wg = (2 * max_cus + 3) / 4;
    const dim3 grid((n_wg + 7) / 8 * 8, n);
    if (rdoq)
      hipLaunchKernelGGL(recon_from_me_multi_kernel<true>, grid, dim3(256), 0, ctx->stream, m,
                         ctx->d_tx_tables, ctx->d_tx_tables_t, lay);
    else
      hipLaunchKernelGGL(recon_from_me_multi_kernel<false>, grid, dim3(256), 0, ctx->stream, m,
                         ctx->d_tx_tables, ctx->d_tx_tables_t, lay);
  }
  if (rdoq) {
    // 3. the quantiser: classification, class lists, the walks
    MultiArgs<RdoqMultiArgs> q;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      xvcgpu_ctx *c = ctxs[i];
      const xvcgpu_status st = ensure_rdoq_scratch(c, a->n_tx, a->n_coeffs);
      if (st != XVCGPU_OK) return st;
      const int cap = c->rdoq_lists_cap;
      RdoqMultiArgs &k = q.a[i];
      k.blocks = a->d_tx;
      k.n = a->n_tx;
      k.coeffs = a->d_coeffs;
      k.d_off = a->d_level_off;
      k.levels = a->d_levels;
      k.nnz_out = a->d_nnz;
      k.l.count = c->d_rdoq_lists;
      for (int cl = 0; cl < 3; cl++) k.l.list[cl] = c->d_rdoq_lists + 4 + (size_t)cl * cap;
      k.l.cls = reinterpret_cast<signed char *>(c->d_rdoq_lists + 4 + 3 * (size_t)cap);
      k.rq_ctx = a->d_rdoq_contexts;
      k.rq_prm = a->d_rdoq_params;
    }
    const int bd = args[0]->rec->bd;
    hipLaunchKernelGGL(rdoq_classify_multi_kernel, dim3((max_tx + 3) / 4, n), dim3(256), 0,
                       ctx->stream, q, bd);
    hipLaunchKernelGGL(rdoq_compact_multi_kernel, dim3(1, n), dim3(1024), 0, ctx->stream, q);
    const int g16 = std::min(max_tx, RDOQ_GRID16),
              g4 = std::min((max_tx + 3) / 4, RDOQ_GRID4), g64 = std::min(max_tx, RDOQ_GRID64);
    hipLaunchKernelGGL(quant_rdo_packed4_multi_kernel, dim3(g16 + g4, n), dim3(64), 0,
                       ctx->stream, q, bd, g16);
    hipLaunchKernelGGL(quant_rdo_packed_multi_kernel, dim3(g64, n), dim3(64), 0,
                       ctx->stream, q, bd);
    // 4. dequantisation + inverse transform + reconstruction
    MultiArgs<InvMultiArgs> v;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      InvMultiArgs &k = v.a[i];
      k.pred = a->scratch_rec->v;
      k.rec = a->scratch_rec->v;
      k.blocks = a->d_tx;
      k.n = a->n_tx;
      k.levels = a->d_levels;
      k.level_off = a->d_level_off;
      k.nnz = a->d_nnz;
    }
    const int n_wg = (max_tx + TX2_WAVES - 1) / TX2_WAVES;
    hipLaunchKernelGGL(inv_wave_multi_kernel, dim3((n_wg + 7) / 8 * 8, n), dim3(64 * TX2_WAVES), 0,
                       ctx->stream, v, ctx->d_tx_tables, ctx->d_tx_tables_t, lay);
    hipLaunchKernelGGL(inv_general_multi_kernel, dim3((max_tx + TX_THREADS - 1) / TX_THREADS, n),
                       dim3(TX_THREADS), 0, ctx->stream, v, ctx->d_tx_tables, lay);
    // 5. the CUs' deblocking records
    MultiArgs<CuInfoMultiArgs> u;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      CuInfoMultiArgs &k = u.a[i];
      k.blocks = a->d_me;
      k.results = a->d_results;
      k.nnz = a->d_nnz;
      k.luma_tx_index = a->d_luma_tx_index;
      k.n = a->n_cus;
      k.qp_y = a->qp_y;
      k.qp_c = a->qp_c;
      k.ref_poc = a->ref_poc;
      k.cus = a->d_cus_own;
    }
    hipLaunchKernelGGL(cu_info_multi_kernel, dim3((max_cus + 255) / 256, n), dim3(256), 0,
                       ctx->stream, u);
  }
  // 6. deblocking, border, SSD parts
  {
    MultiArgs<TailMultiArgs> t;
    int max_tiles = 0;
    for (int i = 0; i < n; i++) {
      const xvcgpu_frame_pass_args *a = args[i];
      xvcgpu_ctx *c = ctxs[i];
      const int tiles = tail_tiles(a->rec->w, a->rec->h);
      const xvcgpu_status st = ensure_tail(c, tiles);
      if (st != XVCGPU_OK) return st;
      if (a->orig->w != a->rec->w || a->orig->h != a->rec->h || !a->d_ssd || !a->d_cus ||
          !a->d_cu_map || a->shift_bitdepth < 8)
        return XVCGPU_INVALID_ARGUMENT;
      TailMultiArgs &k = t.a[i];
      k.d.bd = a->rec->bd;
      k.d.pic_w = a->rec->w;
      k.d.pic_h = a->rec->h;
      k.d.bipred = 0;
      k.d.beta_off = 0;
      k.d.tc_off = 0;
      k.d.sub = 4;
      k.d.y_begin = 0;
      k.d.y_end = a->rec->h;
      k.d.cus = a->d_cus;
      k.d.map = a->d_cu_map;
      k.d.map_stride = a->map_stride;
      k.d.map_rows = (a->rec->h + 3) / 4;
      k.d.comp_mask = 3;
      k.src = a->scratch_rec->v;
      k.dst = a->rec->v;
      k.orig = a->orig->v.c[0];
      k.shift = 2 * (a->shift_bitdepth - 8);
      k.tiles = tiles;
      k.part = c->d_tail_part;
      k.out = reinterpret_cast<unsigned long long *>(a->d_ssd);
      max_tiles = std::max(max_tiles, tiles);
    }
    hipLaunchKernelGGL(deblock_tail_multi_kernel, dim3((max_tiles + 7) / 8 * 8, n), dim3(256), 0,
                       ctx->stream, t);
    hipLaunchKernelGGL(picture_ssd_sum_multi_kernel, dim3(1, n), dim3(256), 0, ctx->stream, t);
  }
  (void)one;
  CHECK_LAUNCH(ctx, "frame_pass_multi");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_affine_me_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *ref,
                                     const xvcgpu_picture *ref_other,
                                     const xvcgpu_affine_me_block *d_blocks, int n,
                                     xvcgpu_affine_me_result *d_results) {
  if (!ctx || !orig || !ref || n < 0 || (n && (!d_blocks || !d_results)))
    return XVCGPU_INVALID_ARGUMENT;
  if (!ref_other) ref_other = ref;
  if (orig->v.bd != ref->v.bd || orig->v.c[0].w != ref->v.c[0].w ||
      orig->v.c[0].h != ref->v.c[0].h || ref_other->v.bd != ref->v.bd ||
      ref_other->v.c[0].w != ref->v.c[0].w || ref_other->v.c[0].h != ref->v.c[0].h)
    return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  // one instance per CU height (a wave per 8 rows of the block)
  hipLaunchKernelGGL(affine_me_kernel<2>, dim3(n), dim3(128), 0, ctx->stream, orig->v.c[0],
                     ref->v.c[0], ref_other->v.c[0], ref->v.bd, d_blocks, n, d_results);
  hipLaunchKernelGGL(affine_me_kernel<4>, dim3(n), dim3(256), 0, ctx->stream, orig->v.c[0],
                     ref->v.c[0], ref_other->v.c[0], ref->v.bd, d_blocks, n, d_results);
  hipLaunchKernelGGL(affine_me_kernel<8>, dim3(n), dim3(512), 0, ctx->stream, orig->v.c[0],
                     ref->v.c[0], ref_other->v.c[0], ref->v.bd, d_blocks, n, d_results);
  CHECK_LAUNCH(ctx, "affine_me_batch");
  return XVCGPU_OK;
}

/* ---- the searches of one CU state into several reference pictures, one launch each ---
 * A CU state's SearchMotion runs the same step for every (list, picture) of the CU:
 * with the single-picture entry points that is one launch per picture, each with one
 * job, one after the other.  Here the pictures come as a table and every job names its
 * slot, so the step is ONE launch whose jobs run side by side; and the caller says
 * which block-size class its jobs are (a CU state's jobs all have the CU's size), so
 * only that class's instances are launched. */
static xvcgpu_status ref_table_of(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                  const xvcgpu_picture *const *refs, int n_refs, RefTable *t) {
  if (!refs || n_refs < 1 || n_refs > XVC_MAX_REF_SLOTS) return XVCGPU_INVALID_ARGUMENT;
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < n_refs; i++) {
    if (!refs[i]) return XVCGPU_INVALID_ARGUMENT;
    if (refs[i]->w != orig->w || refs[i]->h != orig->h || refs[i]->bd != orig->bd)
      return fail(ctx, XVCGPU_INVALID_ARGUMENT, "reference picture mismatch");
    t->pic[i] = refs[i]->v;
  }
  t->n = n_refs;
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_me_search_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                    const xvcgpu_picture *const *refs, int n_refs, int flags,
                                    const xvcgpu_me_block *d_blocks, const uint8_t *d_slots,
                                    int n, xvcgpu_me_result *d_results, int block_class) {
  if (!ctx || !orig || n < 0 || (n && (!d_blocks || !d_slots || !d_results)) ||
      !(flags & (XVCGPU_ME_FULLPEL | XVCGPU_ME_SUBPEL)) || (flags & XVCGPU_ME_LIC_JOBS) ||
      (block_class != 16 && block_class != 32 && block_class != 64))
    return XVCGPU_INVALID_ARGUMENT;
  RefTable t;
  const xvcgpu_status st = ref_table_of(ctx, orig, refs, n_refs, &t);
  if (st != XVCGPU_OK) return st;
  if (n == 0) return XVCGPU_OK;
  Me2Sched sched = {nullptr, nullptr, nullptr};
  if (flags & XVCGPU_ME_FULLPEL) {
    const int e = ctx->me_epoch = (ctx->me_epoch + 1) % 3;
    sched.use = ctx->d_me_rot + e % 3;
    sched.record = ctx->d_me_rot + (e + 1) % 3;
    sched.clear = ctx->d_me_rot + (e + 2) % 3;
  }
#define ME_REFS(MS, PH)                                                                      \
  hipLaunchKernelGGL((me_search_refs_kernel<MS, PH>), me2_grid(n, ME2_WAVES(MS)),            \
                     dim3(64 * ME2_WAVES(MS)), 0, ctx->stream, orig->v, t, d_slots, d_blocks, \
                     n, d_results, ctx->d_tz_pattern, sched, block_class)
#define ME_REFS_SPLIT(MS)                            \
  do {                                               \
    if (flags & XVCGPU_ME_FULLPEL) ME_REFS(MS, 1);   \
    if (flags & XVCGPU_ME_SUBPEL) ME_REFS(MS, 2);    \
  } while (0)
  if (block_class == 16) {
    if ((flags & 3) == 3) ME_REFS(16, 3);
    else ME_REFS_SPLIT(16);
  } else if (block_class == 32) {
    ME_REFS_SPLIT(32);
  } else {
    ME_REFS_SPLIT(64);
    if (flags & XVCGPU_ME_SUBPEL)
      hipLaunchKernelGGL((me_subpel_team_refs_kernel<64, 4>), dim3((n + 7) / 8 * 8), dim3(256), 0,
                         ctx->stream, orig->v, t, d_slots, d_blocks, n, d_results);
  }
#undef ME_REFS_SPLIT
#undef ME_REFS
  CHECK_LAUNCH(ctx, "me_search_refs");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_bipred_search_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                        const xvcgpu_picture *const *refs, int n_refs,
                                        const xvcgpu_bi_block *d_jobs, const uint8_t *d_slots,
                                        int n, xvcgpu_me_result *d_results, int block_class) {
  if (!ctx || !orig || n < 0 || (n && (!d_jobs || !d_slots || !d_results)) ||
      (block_class != 16 && block_class != 32 && block_class != 64))
    return XVCGPU_INVALID_ARGUMENT;
  RefTable t;
  const xvcgpu_status st = ref_table_of(ctx, orig, refs, n_refs, &t);
  if (st != XVCGPU_OK) return st;
  if (n == 0) return XVCGPU_OK;
  const dim3 grid((n + 7) / 8 * 8);
#define BI_REFS(MS)                                                                          \
  hipLaunchKernelGGL((bipred_search_refs_kernel<MS>), grid, dim3(64 * BI_WAVES(MS)), 0,      \
                     ctx->stream, orig->v.c[0], t, d_slots, orig->bd, d_jobs, n, d_results,  \
                     block_class)
  if (block_class == 16) BI_REFS(16);
  else if (block_class == 32) BI_REFS(32);
  else BI_REFS(64);
#undef BI_REFS
  CHECK_LAUNCH(ctx, "bipred_search_refs");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_mc_metric_batch_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *const *refs, int n_refs,
                                          int structural_strength,
                                          const xvcgpu_mc_metric_cand *d_cands,
                                          const uint8_t *d_slots, int n, uint64_t *d_out) {
  if (!ctx || !orig || n < 0 || (n && (!d_cands || !d_slots || !d_out)))
    return XVCGPU_INVALID_ARGUMENT;
  RefTable t;
  const xvcgpu_status st = ref_table_of(ctx, orig, refs, n_refs, &t);
  if (st != XVCGPU_OK) return st;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(mc_metric_refs_kernel, dim3((n + 1) / 2), dim3(128), 0, ctx->stream,
                     orig->v.c[0], t, d_slots, orig->bd, structural_strength, d_cands, n, d_out);
  CHECK_LAUNCH(ctx, "mc_metric_batch_refs");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_affine_me_batch_refs(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                          const xvcgpu_picture *const *refs, int n_refs,
                                          const xvcgpu_affine_me_block *d_blocks,
                                          const uint8_t *d_slots, int n,
                                          xvcgpu_affine_me_result *d_results, int cu_height) {
  if (!ctx || !orig || n < 0 || (n && (!d_blocks || !d_slots || !d_results)) ||
      (cu_height != 16 && cu_height != 32 && cu_height != 64))
    return XVCGPU_INVALID_ARGUMENT;
  RefTable t;
  const xvcgpu_status st = ref_table_of(ctx, orig, refs, n_refs, &t);
  if (st != XVCGPU_OK) return st;
  if (n == 0) return XVCGPU_OK;
#define AFF_REFS(NW)                                                                      \
  hipLaunchKernelGGL(affine_me_refs_kernel<NW>, dim3(n), dim3(64 * NW), 0, ctx->stream,   \
                     orig->v.c[0], t, d_slots, orig->bd, d_blocks, n, d_results)
  if (cu_height == 16) AFF_REFS(2);
  else if (cu_height == 32) AFF_REFS(4);
  else AFF_REFS(8);
#undef AFF_REFS
  CHECK_LAUNCH(ctx, "affine_me_batch_refs");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_copy_segments(xvcgpu_ctx *ctx, const xvcgpu_copy_segment *d_segments,
                                   int n) {
  if (!ctx || n < 0 || (n && !d_segments)) return XVCGPU_INVALID_ARGUMENT;
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(copy_segments_kernel, dim3(32, n), dim3(256), 0, ctx->stream, d_segments,
                     n);
  CHECK_LAUNCH(ctx, "copy_segments");
  return XVCGPU_OK;
}


/* ---- many chains' steps in one launch (k_cs_engine.h) --------------------------- */
struct xvcgpu_cs_env {
  xvcgpu_ctx *ctx;
  CsEnvDev *dev;
};

xvcgpu_status xvcgpu_cs_env_create(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                   const xvcgpu_picture *const *refs, int n_refs,
                                   xvcgpu_picture *s_orig, xvcgpu_picture *s_pred,
                                   xvcgpu_picture *s_rec, int16_t *d_levels,
                                   xvcgpu_cs_result *d_results, xvcgpu_cs_env **out) {
  if (!ctx || !orig || !s_orig || !s_pred || !s_rec || !out) return XVCGPU_INVALID_ARGUMENT;
  if (s_pred->bd != orig->bd || s_rec->bd != orig->bd || s_orig->bd != orig->bd ||
      s_rec->w != s_pred->w || s_rec->h != s_pred->h)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  CsEnvDev h;
  memset(&h, 0, sizeof(h));
  const xvcgpu_status st = ref_table_of(ctx, orig, refs, n_refs, &h.refs);
  if (st != XVCGPU_OK) return st;
  h.orig = orig->v;
  h.s_orig = s_orig->v;
  h.s_pred = s_pred->v;
  h.s_rec = s_rec->v;
  h.levels = d_levels;
  h.results = d_results;
  h.pic_w = orig->w;
  h.pic_h = orig->h;
  xvcgpu_cs_env *e = new (std::nothrow) xvcgpu_cs_env();
  if (!e) return XVCGPU_OUT_OF_MEMORY;
  e->ctx = ctx;
  e->dev = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (hipMalloc(reinterpret_cast<void **>(&e->dev), sizeof(CsEnvDev)) != hipSuccess) {
    delete e;
    return fail(ctx, XVCGPU_OUT_OF_MEMORY, "cs_env");
  }
  if (hipMemcpy(e->dev, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(e->dev);
    delete e;
    return fail(ctx, XVCGPU_DEVICE_ERROR, "cs_env upload");
  }
  *out = e;
  return XVCGPU_OK;
}

void xvcgpu_cs_env_destroy(xvcgpu_cs_env *env) {
  if (!env) return;
  if (env->dev) hipFree(env->dev);
  delete env;
}

#define CS_SEG_RING_HALF (4u << 20)
// room for n segment records in the context's ring (see xvcgpu_internal.h)
static CsSegDev *seg_ring_take(xvcgpu_ctx *ctx, int n) {
  const size_t bytes = (sizeof(CsSegDev) * (size_t)n + 255) & ~(size_t)255;
  if (bytes > CS_SEG_RING_HALF) return nullptr;
  if (!ctx->h_seg_ring) {
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_seg_ring), 2 * CS_SEG_RING_HALF,
                      hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->seg_ring_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->seg_ring_ev[1], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      ctx->h_seg_ring = nullptr;
      return nullptr;
    }
  }
  if (ctx->seg_ring_pos + bytes > CS_SEG_RING_HALF) {
    // this half is full: what read it is behind this event; the other half is free once
    // the launches that read IT have passed
    const int h = ctx->seg_ring_half;
    hipEventRecord(ctx->seg_ring_ev[h], ctx->stream);
    ctx->seg_ring_used[h] = true;
    ctx->seg_ring_half = 1 - h;
    if (ctx->seg_ring_used[1 - h]) hipEventSynchronize(ctx->seg_ring_ev[1 - h]);
    ctx->seg_ring_pos = 0;
  }
  CsSegDev *out = reinterpret_cast<CsSegDev *>(ctx->h_seg_ring + ctx->seg_ring_half * CS_SEG_RING_HALF +
                                              ctx->seg_ring_pos);
  ctx->seg_ring_pos += bytes;
  return out;
}

xvcgpu_status xvcgpu_cs_segs_launch(xvcgpu_ctx *ctx, int kind, const xvcgpu_cs_seg *segs,
                                    int n_segs) {
  if (!ctx || n_segs < 0 || (n_segs && !segs) || kind < 0 || kind >= XVC_CS_SEG_KINDS)
    return XVCGPU_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int kChunk = 16384;          // segments per launch (grid y)
  for (int first = 0; first < n_segs; first += kChunk) {
    const int cnt = std::min(kChunk, n_segs - first);
    CsSegDev *a = seg_ring_take(ctx, cnt);
    if (!a) return fail(ctx, XVCGPU_OUT_OF_MEMORY, "cs_segs_launch: segment ring");
    int max_n = 0, max_nh = 0;
    const int key = segs[first].i0;
    for (int i = 0; i < cnt; i++) {
      const xvcgpu_cs_seg &g = segs[first + i];
      if (g.n < 0 || (kind != XVC_CS_SEG_FETCH && (!g.env || g.env->ctx->device != ctx->device)))
        return XVCGPU_INVALID_ARGUMENT;
      if ((kind == XVC_CS_SEG_ME_REFS || kind == XVC_CS_SEG_BI_REFS ||
           kind == XVC_CS_SEG_AFFINE_REFS) && g.i0 != key)
        return fail(ctx, XVCGPU_INVALID_ARGUMENT, "cs_segs_launch: one kernel instance per call");
      CsSegDev &d = a[i];
      d.n = g.n;
      d.i0 = g.i0;
      d.r0 = g.r0;
      d.r1 = g.r1;
      for (int k = 0; k < 8; k++) d.p[k] = reinterpret_cast<const void *>(g.p[k]);
      d.env = g.env ? g.env->dev : nullptr;
      max_n = std::max(max_n, g.n);
      max_nh = std::max(max_nh, g.n + (g.p[6] ? g.r0 : 0));
    }
    if (max_n == 0) continue;
    const unsigned gy = (unsigned)cnt;
    hipStream_t st = ctx->stream;
    switch (kind) {
      case XVC_CS_SEG_MC_METRIC_REFS:
        hipLaunchKernelGGL(cs_seg_mc_metric_kernel, dim3((max_n + 1) / 2, gy), dim3(128), 0, st, a);
        break;
      case XVC_CS_SEG_START_FOLD:
        hipLaunchKernelGGL(cs_seg_start_fold_kernel, dim3(max_n, gy), dim3(64), 0, st, a);
        break;
      case XVC_CS_SEG_UNI_FOLD:
        hipLaunchKernelGGL(cs_seg_uni_fold_kernel, dim3(max_n, gy), dim3(64), 0, st, a);
        break;
      case XVC_CS_SEG_BI_FOLD:
        hipLaunchKernelGGL(cs_seg_bi_fold_kernel, dim3(max_n, gy), dim3(64), 0, st, a);
        break;
      case XVC_CS_SEG_MERGE_FOLD:
        hipLaunchKernelGGL(cs_seg_merge_fold_kernel, dim3((max_n + 63) / 64, gy), dim3(64), 0, st, a);
        break;
      case XVC_CS_SEG_ME_REFS: {
        const int ep = ctx->me_epoch = (ctx->me_epoch + 1) % 3;
        Me2Sched sched;
        sched.use = ctx->d_me_rot + ep % 3;
        sched.record = ctx->d_me_rot + (ep + 1) % 3;
        sched.clear = ctx->d_me_rot + (ep + 2) % 3;
#define SEG_ME(MS, PH)                                                                      \
  hipLaunchKernelGGL((cs_seg_me_kernel<MS, PH>), dim3(me2_grid(max_n, ME2_WAVES(MS)).x, gy), \
                     dim3(64 * ME2_WAVES(MS)), 0, st, a, ctx->d_tz_pattern, sched)
        if (key == 16) {
          SEG_ME(16, 3);
        } else if (key == 32) {
          SEG_ME(32, 1);
          SEG_ME(32, 2);
        } else if (key == 64) {
          SEG_ME(64, 1);
          SEG_ME(64, 2);
          hipLaunchKernelGGL(cs_seg_me_team_kernel, dim3((max_n + 7) / 8 * 8, gy), dim3(256), 0, st, a);
        } else {
          return XVCGPU_INVALID_ARGUMENT;
        }
#undef SEG_ME
        break;
      }
      case XVC_CS_SEG_BI_REFS: {
        const dim3 grid((max_n + 7) / 8 * 8, gy);
        if (key == 16) hipLaunchKernelGGL(cs_seg_bi_kernel<16>, grid, dim3(64 * BI_WAVES(16)), 0, st, a);
        else if (key == 32) hipLaunchKernelGGL(cs_seg_bi_kernel<32>, grid, dim3(64 * BI_WAVES(32)), 0, st, a);
        else if (key == 64) hipLaunchKernelGGL(cs_seg_bi_kernel<64>, grid, dim3(64 * BI_WAVES(64)), 0, st, a);
        else return XVCGPU_INVALID_ARGUMENT;
        break;
      }
      case XVC_CS_SEG_AFFINE_REFS:
        if (key == 16) hipLaunchKernelGGL(cs_seg_affine_kernel<2>, dim3(max_n, gy), dim3(128), 0, st, a);
        else if (key == 32) hipLaunchKernelGGL(cs_seg_affine_kernel<4>, dim3(max_n, gy), dim3(256), 0, st, a);
        else if (key == 64) hipLaunchKernelGGL(cs_seg_affine_kernel<8>, dim3(max_n, gy), dim3(512), 0, st, a);
        else return XVCGPU_INVALID_ARGUMENT;
        break;
      case XVC_CS_SEG_INTER_PRED:
        hipLaunchKernelGGL(cs_seg_inter_pred_kernel, dim3(max_n, gy), dim3(256), 0, st, a);
        break;
      case XVC_CS_SEG_RESIDUAL_AT:
        // as xvcgpu_residual_rdoq_batch_at checks a single call: <= 64 blocks, the arrays
        // of the blocks, 0 ... 64 head candidates, candidates and their output both or neither
        for (int i = 0; i < cnt; i++) {
          const xvcgpu_cs_seg &g = segs[first + i];
          if (g.n > 64 || !g.p[5] || g.r0 < 0 || g.r0 > 64 || (g.p[6] != 0) != (g.p[7] != 0) ||
              (g.n && (!g.p[0] || !g.p[1] || !g.p[2] || !g.p[3] || !g.p[4])))
            return fail(ctx, XVCGPU_INVALID_ARGUMENT, "cs_segs_launch: residual segment");
        }
        hipLaunchKernelGGL(cs_seg_residual_kernel, dim3(max_nh, gy), dim3(TX_THREADS), 0, st, a,
                           ctx->d_tx_tables, ctx->d_tx_tables_t, xvcgpu_tx_layout());
        break;
      case XVC_CS_SEG_EVAL_DIST:
        hipLaunchKernelGGL(cs_seg_eval_dist_kernel, dim3((max_n + 3) / 4, gy), dim3(256), 0, st, a);
        break;
      case XVC_CS_SEG_FETCH:
        for (int i = 0; i < cnt; i++)
          if ((segs[first + i].n & 3) || ((segs[first + i].p[0] | segs[first + i].p[1]) & 3))
            return XVCGPU_INVALID_ARGUMENT;
        hipLaunchKernelGGL(cs_seg_fetch_kernel, dim3(8, gy), dim3(256), 0, st, a);
        break;
      default:
        return XVCGPU_INVALID_ARGUMENT;
    }
    CHECK_LAUNCH(ctx, "cs_segs_launch");
  }
  return XVCGPU_OK;
}

/* ---- the folds of one SearchMotion chain (k_cu_state.h) ----------------------- */
xvcgpu_status xvcgpu_cs_start_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                   const uint64_t *d_start_dist, xvcgpu_me_block *d_me_jobs,
                                   const xvcgpu_me_result *d_me_res,
                                   xvcgpu_affine_me_block *d_aff_jobs,
                                   xvcgpu_cs_result *d_results, int pic_w, int pic_h) {
  if (!ctx || n < 0 || first < 0 || (n && (!d_passes || !d_start_dist || !d_results)))
    return XVCGPU_INVALID_ARGUMENT;
  if (!n) return XVCGPU_OK;
  hipLaunchKernelGGL(cs_start_fold_kernel, dim3(n), dim3(64), 0, ctx->stream, d_passes,
                     first, n, d_start_dist, d_me_jobs, d_me_res, d_aff_jobs, d_results, pic_w, pic_h);
  CHECK_LAUNCH(ctx, "cs_start_fold");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_cs_merge_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_merge *d_merges, int first,
                                   int n, const uint64_t *d_dist,
                                   const xvcgpu_inter_block *d_cands,
                                   xvcgpu_cs_merge_result *d_results,
                                   xvcgpu_inter_block *d_ev_inter) {
  if (!ctx || n < 0 || first < 0 || (n && (!d_merges || !d_dist || !d_cands || !d_results)))
    return XVCGPU_INVALID_ARGUMENT;
  if (!n) return XVCGPU_OK;
  hipLaunchKernelGGL(cs_merge_fold_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream,
                     d_merges, first, n, d_dist, d_cands, d_results, d_ev_inter);
  CHECK_LAUNCH(ctx, "cs_merge_fold");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_cs_uni_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                 const xvcgpu_me_result *d_me_res,
                                 const xvcgpu_affine_me_result *d_aff_res,
                                 xvcgpu_cs_result *d_results, xvcgpu_bi_block *d_bi_jobs,
                                 xvcgpu_affine_me_block *d_aff_jobs) {
  if (!ctx || n < 0 || (n && (!d_passes || !d_results))) return XVCGPU_INVALID_ARGUMENT;
  if (!n) return XVCGPU_OK;
  hipLaunchKernelGGL(cs_uni_fold_kernel, dim3(n), dim3(64), 0, ctx->stream, d_passes, first, n,
                     d_me_res, d_aff_res, d_results, d_bi_jobs, d_aff_jobs);
  CHECK_LAUNCH(ctx, "cs_uni_fold");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_cs_bi_fold(xvcgpu_ctx *ctx, const xvcgpu_cs_pass *d_passes, int first,
    int n,
                                const xvcgpu_me_result *d_bi_res,
                                const xvcgpu_affine_me_result *d_aff_res,
                                xvcgpu_cs_result *d_results, xvcgpu_inter_block *d_ev_inter) {
  if (!ctx || n < 0 || (n && (!d_passes || !d_results))) return XVCGPU_INVALID_ARGUMENT;
  if (!n) return XVCGPU_OK;
  hipLaunchKernelGGL(cs_bi_fold_kernel, dim3(n), dim3(64), 0, ctx->stream, d_passes, first, n,
                     d_bi_res, d_aff_res, d_results, d_ev_inter);
  CHECK_LAUNCH(ctx, "cs_bi_fold");
  return XVCGPU_OK;
}

xvcgpu_status xvcgpu_eval_dist_batch(xvcgpu_ctx *ctx, const xvcgpu_picture *orig,
                                     const xvcgpu_picture *pred, const xvcgpu_picture *rec,
                                     int structural_strength, const xvcgpu_eval_cand *d_cands,
                                     int n, uint64_t *d_out) {
  if (!ctx || !orig || !pred || !rec || n < 0 || (n && (!d_cands || !d_out)))
    return XVCGPU_INVALID_ARGUMENT;
  if (orig->bd != pred->bd || orig->bd != rec->bd)
    return fail(ctx, XVCGPU_INVALID_ARGUMENT, "picture mismatch");
  if (n == 0) return XVCGPU_OK;
  hipLaunchKernelGGL(eval_dist_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, orig->v,
                     pred->v, rec->v, structural_strength, d_cands, n, d_out);
  CHECK_LAUNCH(ctx, "eval_dist_batch");
  return XVCGPU_OK;
}

}  // extern "C"
