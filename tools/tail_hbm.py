#!/usr/bin/env python3
"""The fused tail of the frame pass (xvcgpu_deblock_pad_ssd: deblocking of both
edge directions, PadBorder, PSNR parts in one launch) as an HBM measurement: one
1080p picture (12 MB) lives in the 256 MB Infinity Cache, so the launch is timed
over N DISTINCT picture triples (source, destination, original) visited in turn -
N x 26 MB of pictures, far more than the cache holds - and the algorithmic bytes
(2 x 1.5 N S + N S + border) divided by the average launch time.
Run on the GPU box:  python tools/tail_hbm.py [N=40] [width height]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xvc_amd import api, pipeline, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bd = 10
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
fp = pipeline.FramePass(ctx, W, H, bd, qp=32, rdoq=True)
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), 128)
O.upload(pad(clip.frame(1)), 128)
fp.run(O, R, Rec)          # CU records + map of a real pass
ctx.sync()
d = fp.desc
trip = []
for i in range(N):
    s, t, o = (ctx.picture(W, H, bd) for _ in range(3))
    s.upload(pad(clip.frame(i % 7 + 1)), 128)     # an unfiltered "reconstruction"
    o.upload(pad(clip.frame((i + 1) % 7 + 1)), 128)
    trip.append((s, t, o))
ctx.sync()


def launch(s, t, o):
    ctx.deblock_pad_ssd_dev(s, t, o, fp.d_cus.ptr, d.n_cus_total, fp.d_map.ptr,
                            d.cu_map.shape[1], 0, 0, 0, bd, fp.d_ssd.ptr)


for s, t, o in trip:
    launch(s, t, o)
ctx.sync()
rounds = 5
ctx.timer_begin()
for _ in range(rounds):
    for s, t, o in trip:
        launch(s, t, o)
ms = ctx.timer_end() / (rounds * N)
ctx.timer_begin()
for _ in range(rounds * N):
    launch(*trip[0])
ms_one = ctx.timer_end() / (rounds * N)
S = 2
luma = W * H
border = 2 * 128 * (W + H + 256) * S * 1.5
alg = 2 * 1.5 * luma * S + luma * S + border
pic_mb = 3 * ctx.lib.xvcgpu_picture_bytes(W, H) / 1e6
print("deblock_pad_ssd, %dx%d, back to back on one stream (launch + SSD fold):" % (W, H))
print("  %d distinct picture triples (%.0f MB of pictures): %.1f us per launch, %.0f GB/s "
      "algorithmic (%.1f MB per launch) = %.1f %% of 8 TB/s, %.1f %% of the ~6.3 TB/s a "
      "copy reaches" % (N, N * pic_mb, 1e3 * ms, alg / ms / 1e6, alg / 1e6,
                        100 * alg / ms / 1e6 / 8000, 100 * alg / ms / 1e6 / 6300))
print("  one triple over and over (cache resident):            %.1f us per launch, %.0f GB/s"
      % (1e3 * ms_one, alg / ms_one / 1e6))
