"""Times the motion search per block-size class on 1080p content (run on the GPU
box): every CU of one size per launch, 8x8 ... 64x64 and the non-square shapes of
binary splits; reports us per launch and ns per luma sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from xvc_amd import api, pipeline, synth

W, H, bd, border = 1920, 1080, int(os.environ.get("BD", "10")), 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
lam = pipeline.lambda16_for_qp(32)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.timer_begin()
    for _ in range(reps): fn()
    return ctx.timer_end() / reps
for (w, h) in [(8, 8), (16, 16), (16, 8), (8, 16), (32, 32), (32, 16), (16, 32), (32, 8), (64, 64), (64, 32), (32, 64), (64, 16)]:
    xs, ys = np.meshgrid(np.arange(0, W - w + 1, w), np.arange(0, H - h + 1, h))
    n = xs.size
    me = np.zeros(n, api.ME_DTYPE)
    me["x"], me["y"], me["w"], me["h"] = xs.ravel(), ys.ravel(), w, h
    me["depth_nonzero"], me["lambda16"], me["search_range"] = 1, lam, 96
    d_me, d_res = ctx.buffer(me), ctx.alloc(api.MERES_DTYPE.itemsize * n)
    ms = max(w, h)
    out = []
    for flags in (1, 2, 3):
        out.append(1e3 * timed(lambda: ctx.me_search_dev(O, R, flags, d_me.ptr, n, d_res.ptr, ms)))
    print("%2dx%-2d %5d CUs: full-pel %7.1f us, sub-pel %7.1f us, fused %7.1f us = %.3f ns/sample" %
          (w, h, n, out[0], out[1], out[2], 1e3 * out[2] / (n * w * h)))
    d_me.free(); d_res.free()
