"""Timing of the all-intra picture pass (pipeline.IntraPicturePass) at 1080p:
encoder side (with the per-wave host arg-min) and decoder side (plain launches
and as one recorded HIP graph)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, qp = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 10, 32
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
O, R, D = (ctx.picture(W, H, bd) for _ in range(3))
O.upload(clip.frame(3))
ip = pipeline.IntraPicturePass(ctx, W, H, bd, qp, 16)
ip.encode(O, R)
t = time.perf_counter(); ip.encode(O, R, host_select=True); ctx.sync(); th = time.perf_counter() - t
t = time.perf_counter(); ip.encode(O, R); ctx.sync(); te = time.perf_counter() - t
modes, levels, nnz = ip.results()
ip.load(modes, levels, nnz)
ip.decode(D); ctx.sync()
t = time.perf_counter()
for _ in range(5): ip.decode(D)
ctx.sync(); td = (time.perf_counter() - t) / 5
rec = ctx.record(lambda: ip.decode(D))
ctx.replay(rec); ctx.sync()
t = time.perf_counter()
for _ in range(5): ctx.replay(rec)
ctx.sync(); tg = (time.perf_counter() - t) / 5
print("waves %d  encode %.2f ms (host fold %.2f ms)  decode %.2f ms  decode as graph %.2f ms" %
      (len(list(ip.desc.waves())), te * 1e3, th * 1e3, td * 1e3, tg * 1e3))
