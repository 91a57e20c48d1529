import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R, Rec = ctx.picture(W, H, bd), ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd, rdoq=os.environ.get("QUANT", "rdoq") == "rdoq")
for _ in range(5): fp.run(O, R, Rec)
ctx.sync()
N = int(os.environ.get("N", "30"))
t0 = time.perf_counter()
for _ in range(N): fp.run(O, R, Rec)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print("host enqueue per step %.1f us, total per step %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
