import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
for cu in (16, 8):
    O, R, Rec = ctx.picture(W, H, bd), ctx.picture(W, H, bd), ctx.picture(W, H, bd)
    R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
    fp = pipeline.FramePass(ctx, W, H, bd, qp=32, cu=cu, rdoq=True)
    fp.run(O, R, Rec); ctx.sync()
    out = {}
    for name, fn in fp.kernel_steps(O, R, Rec, ref_poc=0):
        fn(); ctx.sync(); ctx.timer_begin()
        for _ in range(20): fn()
        out[name] = round(ctx.timer_end() / 20 * 1e3, 1)
    print("cu", cu, "n_cus", fp.desc.n_cus, out)
