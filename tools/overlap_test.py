#!/usr/bin/env python3
"""How much does one kernel of the frame pass slow another one down when they
run from two streams at once?  (run on the GPU box)  For each pair (A, B):
B alone, then B while stream A keeps re-running its kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = 1920, 1080, 10
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
ctxs = [api.Context(0), api.Context(0), api.Context(0)]
state = []
for c in ctxs:
    O, R, Rec = (c.picture(W, H, bd) for _ in range(3))
    R.upload(pad(clip.frame(0)), 128)
    O.upload(pad(clip.frame(1)), 128)
    fp = pipeline.FramePass(c, W, H, bd, qp=32, rdoq=True)
    fp.run(O, R, Rec)
    c.sync()
    state.append((fp, dict(fp.kernel_steps(O, R, Rec))))


def rate(c, fn, n):
    c.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    c.sync()
    return (time.perf_counter() - t0) / n * 1e6


names = ["me_search", "fwd_from_me", "quant_rdo", "inv_transform", "deblock_pad_ssd"]
for b in names:
    fb = state[1][1][b]
    alone = rate(ctxs[1], fb, 200)
    out = ["%-14s alone %6.1f us |" % (b, alone)]
    for a in names:
        fa = state[0][1][a]
        # keep stream A busy: enqueue plenty of A first, then time B
        for _ in range(400):
            fa()
        t = rate(ctxs[1], fb, 100)
        ctxs[0].sync()
        out.append("%s: %6.1f" % (a[:9], t))
    print(" ".join(out))

# two background streams
print("two background streams (A, A2) vs B")
for b in ["me_search", "fwd_from_me", "inv_transform", "quant_rdo"]:
    fb = state[1][1][b]
    alone = rate(ctxs[1], fb, 200)
    out = ["%-14s alone %6.1f us |" % (b, alone)]
    for a, a2 in [("quant_rdo", "quant_rdo"), ("quant_rdo", "me_search"), ("me_search", "me_search")]:
        for _ in range(400):
            state[0][1][a]()
            state[2][1][a2]()
        t = rate(ctxs[1], fb, 100)
        ctxs[0].sync()
        ctxs[2].sync()
        out.append("%s+%s: %6.1f" % (a[:5], a2[:5], t))
    print(" ".join(out))
