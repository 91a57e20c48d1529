#!/usr/bin/env python3
"""The fused tail of the frame pass batched over the pictures of B chains in ONE launch
(deblock_tail_multi_kernel of xvcgpu_frame_pass_multi, grid y = picture), over G groups of
distinct pictures visited in turn so that no launch finds its pictures in the Infinity
Cache (one group's pass at a time: the host waits between them).  Run under rocprofv3 --kernel-trace --stats (tools/tail_batched.sh), which reads the
tail kernel's average duration and prices it against SURVEY 8d's bytes
(2 x 1.5 N S + N S + border per picture) x B.
    python tools/tail_batched.py [B=4] [G=4] [width height] [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from xvc_amd import api, pipeline, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 10
bd = 10
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
frames = [pad(clip.frame(k)) for k in range(3)]
groups = []
for g in range(G):
    ctxs = [api.Context(0) for _ in range(B)]
    passes = [pipeline.FramePass(c, W, H, bd, qp=32, rdoq=True) for c in ctxs]
    pipeline.share_stream(passes)
    origs, refs, recs = [], [], []
    for i, c in enumerate(ctxs):
        o, r, t = (c.picture(W, H, bd) for _ in range(3))
        o.upload(frames[1 + (i + g) % 2], 128)
        r.upload(frames[0], 128)
        origs.append(o); refs.append(r); recs.append(t)
    groups.append((ctxs, passes, origs, refs, recs))
for _ in range(rounds):
    for ctxs, passes, origs, refs, recs in groups:
        pipeline.run_multi(passes, origs, refs, recs, [0] * B)
        ctxs[0].sync()      # the groups have their own streams: one pass on the device at a time
S = 2
luma = W * H
border = 2 * 128 * (W + H + 256) * S * 1.5
alg = 2 * 1.5 * luma * S + luma * S + border
print("TAIL_BATCHED B=%d G=%d %dx%d launches=%d alg_bytes_per_picture=%d" % (B, G, W, H, rounds * G, alg))
