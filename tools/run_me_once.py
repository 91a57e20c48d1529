"""Runs the ME kernel a few times on the 1080p bench workload (for rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
# XCD_TILES=1: the CU list region-major (bench.py's default order)
fp = pipeline.FramePass(ctx, W, H, bd, xcd_tiles=os.environ.get("XCD_TILES", "0") == "1")
d = fp.desc
if flags != 3:  # real full-pel results for a sub-pel-only run
    ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
for _ in range(5):
    ctx.me_search_dev(O, R, flags, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
ctx.sync()
