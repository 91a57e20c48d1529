"""Are per-job ME durations predictable from the previous picture? (XVCGPU_TRACE build)"""
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
R.upload(pad(clip.frame(0)), border)
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
lib = api.load_library()
lifes = []
for n in range(1, 7):
    O.upload(pad(clip.frame(n)), border)
    ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16); ctx.sync()
    buf = np.zeros((d.n_cus, 16), np.uint64)
    lib.xvcgpu_debug_me_trace(buf.ctypes.data_as(C.c_void_p), d.n_cus)
    t = buf.astype(np.int64)
    lifes.append(t[:, 8] - t[:, 0])
    fp.run(O, R, Rec); ctx.sync()   # (the recon kernel reuses the trace rows)
    R, Rec = Rec, R
for a, b in zip(lifes[1:-1], lifes[2:]):
    print("corr(job duration, next picture) = %.3f; mean %.0f std %.0f" % (np.corrcoef(a, b)[0, 1], a.mean(), a.std()))
# how good is "longest first" with last picture's durations as the predictor?
def makespan(order, dur, slots=4096):
    import heapq
    h = [0.0] * slots
    heapq.heapify(h)
    for j in order:
        heapq.heappush(h, heapq.heappop(h) + dur[j])
    return max(h)
a, b = lifes[-2].astype(float), lifes[-1].astype(float)
n = len(b)
print("greedy in index order: %.0f   LPT by previous durations: %.0f   LPT oracle: %.0f   lower bound: %.0f" %
      (makespan(range(n), b), makespan(np.argsort(-a), b), makespan(np.argsort(-b), b), b.sum() / 4096))
