import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
for rep in (1, 2, 3, 4):
    me = np.concatenate([d.me] * rep)
    dm = ctx.buffer(me)
    dr = ctx.alloc(api.MERES_DTYPE.itemsize * len(me))
    for _ in range(3):
        ctx.me_search_dev(O, R, 3, dm.ptr, len(me), dr.ptr, 16)
    ctx.sync(); ctx.timer_begin()
    for _ in range(50):
        ctx.me_search_dev(O, R, 3, dm.ptr, len(me), dr.ptr, 16)
    ms = ctx.timer_end() / 50
    print("jobs x%d: %.1f us per launch, %.1f us per picture" % (rep, ms * 1e3, ms * 1e3 / rep))
