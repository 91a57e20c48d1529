"""Times the ME kernel phases separately on the 1080p bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from xvc_amd import api, pipeline, synth

W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd, square=not os.environ.get('NOSQUARE'))
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
def timed(fn, reps=20):
    fn(); ctx.sync(); ctx.timer_begin()
    for _ in range(reps): fn()
    return ctx.timer_end() / reps
for name, flags in (("fullpel", 1), ("subpel", 2), ("both", 3), ("both, 16 class only", 3),
                    ("both, 16 class only, sq16 hint", 3 | api.ME_HINT_SQ16)):
    ms = 16 if "16 class" in name else 64
    t = timed(lambda: ctx.me_search_dev(O, R, flags, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, ms))
    print(name, "%.4f ms" % t)
res = fp.d_res.to_array(api.MERES_DTYPE, d.n_cus)
print("mv median", np.median(res["mv_x"]), np.median(res["mv_y"]))
