"""Section timeline of the RDOQ walk (developer build, -DXVCGPU_TRACE):

    hipcc ... -DXVCGPU_TRACE -o xvc_amd/libxvcgpu_trace.so      (tools/trace_rdoq.sh)
    XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so python tools/trace_rdoq.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd, qp=32, rdoq=True)
# CHAIN=n: the state of a chain after n pictures (the bench's steady state is
# reached after a few hundred: tools/chain_drift.py); default: a chain's first pictures
chain = int(os.environ.get("CHAIN", "0"))
steady = os.environ.get("STATE", "") == "steady"   # the chain of tools/run_rdoq_steady.py / throughput_cost.py
for j in range(chain):
    k = j % 14
    O.upload(pad(clip.frame(j % 7 + 1 if steady else (k if k < 8 else 14 - k))), border)
    fp.run(O, R, Rec)
    R, Rec = Rec, R
if steady:
    O.upload(pad(clip.frame(chain % 7 + 1)), border)
for _ in range(3):
    fp.run(O, R, Rec)
ctx.sync()
lib = api.load_library()
counts = (C.c_int32 * 3)()
lib.xvcgpu_quant_rdo_class_counts(ctx.h, counts)
rows = min(2048, counts[1])   # a block per workgroup (k_rdoq4.h)
buf = np.zeros((4096, 16), np.uint64)
lib.xvcgpu_debug_rdoq_trace(buf.ctypes.data_as(C.c_void_p), 4096)
steps = buf[:rows, 11:15].astype(np.int64)
steps2 = buf[2048:2048 + rows, 11:15].astype(np.int64)
t = buf[:rows, :11].astype(np.int64)
ok = (t[:, 10] > t[:, 0]) & (t[:, 4] > 0)
t = t[ok]
steps = steps[ok]
steps2 = steps2[ok]
print("%d waves of the class of up to sixteen sub-blocks, a block each (%d blocks); clock ticks" % (len(t), counts[1]))
names = ["count,list", "block,prm,off", "coefficients", "ctx costs", "quant+last", "setup",
         "diagonals", "EvalLastPos", "zero+signs", "sign hide", "levels out"]
life = t[:, 10] - t[:, 0]
rt = buf[2048:2048 + rows, :2].astype(np.int64)[ok]
rt_us = (rt[:, 1] - rt[:, 0]) / 100.0          # s_memrealtime: 100 MHz
good = (rt_us > 0) & (rt[:, 0] > 0) & (rt_us < 1e6)   # (rows the kernel did not stamp)
print("wall clock per wave life: mean %.1f us p50 %.1f p90 %.1f max %.1f; s_memtime ticks per us: %.0f" %
      (rt_us[good].mean(), np.median(rt_us[good]), np.percentile(rt_us[good], 90), rt_us[good].max(),
       (life[good] / rt_us[good]).mean()))
g = good
st0 = rt[g, 0].min()
print("wall clock, relative to the first wave's start: wave starts p50 %.1f p90 %.1f max %.1f us; "
      "wave ends p50 %.1f p90 %.1f max %.1f us" %
      (np.median(rt[g, 0] - st0) / 100.0, np.percentile(rt[g, 0] - st0, 90) / 100.0,
       (rt[g, 0] - st0).max() / 100.0, np.median(rt[g, 1] - st0) / 100.0,
       np.percentile(rt[g, 1] - st0, 90) / 100.0, (rt[g, 1] - st0).max() / 100.0))
print("wave lifetime: mean %.0f p50 %.0f p90 %.0f max %d" % (life.mean(), np.median(life), np.percentile(life, 90), life.max()))
prev = t[:, 0]
for k in range(1, 11):
    # a section the wave skipped keeps the stamp of an earlier block of the same
    # workgroup (or none): only stamps inside [previous section, end] count
    cur = np.where((t[:, k] >= prev) & (t[:, k] <= t[:, 10]), t[:, k], prev)
    d = cur - prev
    print("%-14s mean %7.0f  p50 %7.0f  p90 %7.0f  max %8d  share %5.1f%%" %
          (names[k], d.mean(), np.median(d), np.percentile(d, 90), d.max(), 100.0 * d.sum() / life.sum()))
    prev = cur
print("inside the diagonal loop (sums over the diagonals): step 1 (decisions) %.0f, step 2 (no choice) %.0f, "
      "step 3 (zero sub-block) %.0f, loop head %.0f" % tuple(steps.mean(axis=0)[[0, 1, 2, 3]]))
print("inside EvalLastPos: own sums %.0f, stop + table %.0f, prefix + candidates %.0f, reduce %.0f" %
      tuple(steps2.mean(axis=0)))
