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
for flags in (1, 2, 3):
    for _ in range(5):
        ctx.me_search_dev(O, R, flags, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
    ctx.sync()
