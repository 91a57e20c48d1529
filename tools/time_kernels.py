"""Isolated timings of the frame-pass kernels on the 1080p bench workload."""
import os, sys
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
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
fp.run(O, R, Rec); ctx.sync()
def timed(fn, reps=50):
    fn(); ctx.sync(); ctx.timer_begin()
    for _ in range(reps): fn()
    return 1e3 * ctx.timer_end() / reps
which = sys.argv[1:] or ["me", "recon", "deblock", "pad", "ssd"]
fns = {
    "me": lambda: ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16),
    "recon": lambda: ctx.recon_from_me_dev(O, R, Rec, fp.d_me.ptr, fp.d_res.ptr, d.n_cus, d.qp, d.qp_c, 0, fp.d_nnz.ptr, fp.d_cus_own),
    "deblock": lambda: ctx.deblock_dev(Rec, fp.d_cus.ptr, d.n_cus_total, fp.d_map.ptr, d.cu_map.shape[1]),
    "pad": lambda: ctx.pad_border(Rec),
    "ssd": lambda: ctx.picture_ssd_dev(O, Rec, 0, bd, fp.d_ssd.ptr),
}
for k in which:
    print("%-8s %7.2f us" % (k, timed(fns[k])))
