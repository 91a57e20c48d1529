"""Per-job phase timeline of the ME kernel (developer build, -DXVCGPU_TRACE):

    hipcc ... -DXVCGPU_TRACE -o xvc_amd/libxvcgpu_trace.so
    XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so python tools/trace_me.py [flags]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
lib = api.load_library()
chain = int(os.environ.get("CHAIN", "0"))   # run this many chained frame passes first
if chain:
    Rec = ctx.picture(W, H, bd)
    for n_ in range(1, chain + 1):
        k = (n_ - 1) % 14
        O.upload(pad(clip.frame(1 + (k if k < 8 else 14 - k))), border)
        fp.run(O, R, Rec)
        R, Rec = Rec, R
    k = chain % 14
    O.upload(pad(clip.frame(1 + (k if k < 8 else 14 - k))), border)
for _ in range(3):
    ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
ctx.sync(); ctx.timer_begin()
ctx.me_search_dev(O, R, flags, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
ms = ctx.timer_end()
n = d.n_cus
buf = np.zeros((n, 24), np.uint64)
lib.xvcgpu_debug_me_trace(buf.ctypes.data_as(C.c_void_p), n)
t = buf[:, :9].astype(np.int64)
t0 = t[:, 0].min()
names = ["desc+orig", "predictors", "raster", "neighbour", "grid", "refine", "subpel win", "subpel"]
print("kernel %.4f ms; span of timestamps %d ticks -> %.1f ticks/us" % (ms, t[:, 8].max() - t0, (t[:, 8].max() - t0) / (ms * 1e3)))
life = t[:, 8] - t[:, 0]
print("wave lifetime ticks: mean %.0f p50 %.0f p90 %.0f max %d" % (life.mean(), np.median(life), np.percentile(life, 90), life.max()))
for k, nm in enumerate(names):
    dph = t[:, k + 1] - t[:, k]
    print("%-12s mean %7.0f  p50 %7.0f  p90 %7.0f  max %8d  share %5.1f%%" % (nm, dph.mean(), np.median(dph), np.percentile(dph, 90), dph.max(), 100.0 * dph.sum() / life.sum()))
start = t[:, 0] - t0
print("start time ticks: p10 %.0f p50 %.0f p90 %.0f max %d" % (np.percentile(start, 10), np.median(start), np.percentile(start, 90), start.max()))

# ---- concurrency over time (s_memrealtime: 100 MHz, device-wide) ----
rt = buf[:, 9:11].astype(np.int64)
ok = (rt[:, 0] > 0) & (rt[:, 1] >= rt[:, 0])
rt = rt[ok]
a0 = rt[:, 0].min()
st, en = (rt[:, 0] - a0) / 100.0, (rt[:, 1] - a0) / 100.0   # microseconds
span = en.max()
print("wall span %.1f us; wave lifetime mean %.1f us p90 %.1f max %.1f" %
      (span, (en - st).mean(), np.percentile(en - st, 90), (en - st).max()))
print("active waves every 5 us:", [int(((st <= x) & (en > x)).sum()) for x in np.arange(2.5, span, 5.0)])
print("starts per 5 us:        ", np.histogram(st, bins=np.arange(0, span + 5, 5.0))[0].tolist())
late = np.argsort(en)[-5:]
grid = (t[:, 5] - t[:, 4]) > 4 * np.median(t[:, 5] - t[:, 4]) + 2000
print("jobs that ran the step-5 grid:", int(grid.sum()))
print("last 5 waves: start/end us", [(round(float(st[i]), 1), round(float(en[i]), 1)) for i in late])

cnt = buf[:, 11:16].astype(np.int64)
clk = buf[:, 16:24].astype(np.int64)
for k, nm in enumerate(["diamond sweeps", "16-cand passes", "refine iters", "neighbour steps", "candidates"]):
    v = cnt[:, k]
    print("%-16s mean %6.2f p50 %4d p90 %4d max %5d" % (nm, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
print("sweeps histogram", np.bincount(cnt[:, 0].clip(0, 12)).tolist())
print("refine histogram", np.bincount(cnt[:, 2].clip(0, 12)).tolist())
for k, nm in enumerate(["sp setup", "sp planes", "sp records", "sp sweep"]):
    v = clk[:, k]
    print("%-12s mean %7.0f  p50 %7.0f  p90 %7.0f" % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
