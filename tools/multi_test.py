#!/usr/bin/env python3
"""Frame passes/s with n pictures per launch (xvcgpu_frame_pass_multi), chained
round to round like bench.py's chains.  (run on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = 1920, 1080, 10
N = int(os.environ.get("STEPS", 300))
rdoq = os.environ.get("QUANT", "rdoq") == "rdoq"
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
F = 8
def make_group(K):
    ctxs = [api.Context(0) for _ in range(K)]
    passes = [pipeline.FramePass(c, W, H, bd, qp=32, rdoq=rdoq) for c in ctxs]
    pipeline.share_stream(passes)
    origs = [[c.picture(W, H, bd) for _ in range(F)] for c in ctxs]
    for i, c in enumerate(ctxs):
        for k in range(F):
            origs[i][k].upload(pad(clip.frame(k + 1)), 128)
    recs = [[c.picture(W, H, bd), c.picture(W, H, bd)] for c in ctxs]
    for i in range(K):
        recs[i][0].upload(pad(clip.frame(0)), 128)
    return ctxs, passes, origs, recs


def frame(j, i):
    k = (j + 2 * i) % (2 * F - 2)
    return k if k < F else 2 * F - 2 - k


for G, K in ((1, 3), (2, 2), (2, 3), (3, 2), (3, 3), (2, 4), (4, 2)):
    groups = [make_group(K) for _ in range(G)]

    def run(n):
        for g in groups:
            g[0][0].sync()
        t0 = time.perf_counter()
        for j in range(n):
            for gi, (ctxs, passes, origs, recs) in enumerate(groups):
                pipeline.run_multi(passes, [origs[i][frame(j + gi, i)] for i in range(K)],
                                   [recs[i][j & 1] for i in range(K)],
                                   [recs[i][(j + 1) & 1] for i in range(K)], [j] * K)
        for g in groups:
            g[0][0].sync()
        return n * K * G / (time.perf_counter() - t0)

    run(20)
    print("%d streams x %d pictures per launch: %.0f / %.0f passes/s" % (G, K, run(N), run(N)))
    for ctxs, passes, origs, recs in groups:
        for p in passes:
            p.destroy()
        for c in ctxs:
            c.close()
