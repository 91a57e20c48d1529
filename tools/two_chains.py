"""Throughput of K independent frame-pass chains issued round-robin on K
streams of one GPU (pictures of different chains overlap each other's drains)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 10, 128
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
clip = synth.SyntheticClip(W, H, bd)
F = 8
for K in [int(v) for v in os.environ.get("CHAINS", "1,2,3").split(",")]:
    ctxs = [api.Context(0) for _ in range(K)]
    if os.environ.get("PRIO") == "1" and K >= 2:   # stagger: first chain high, last low
        ctxs[0].use_priority_stream(True)
        ctxs[-1].use_priority_stream(False)
    chains = []
    for k, ctx in enumerate(ctxs):
        origs = []
        for n in range(1, F + 1):
            p = ctx.picture(W, H, bd); p.upload(pad(clip.frame(n + 3 * k)), border); origs.append(p)
        recs = [ctx.picture(W, H, bd), ctx.picture(W, H, bd)]
        recs[0].upload(pad(clip.frame(3 * k)), border)
        chains.append((ctx, pipeline.FramePass(ctx, W, H, bd, qp=32), origs, recs))
    GRAPH = os.environ.get("GRAPH") == "1"
    graphs = {}
    def step(i):
        for ci, (ctx, fp, origs, recs) in enumerate(chains):
            k = i % (2 * F - 2)
            o = origs[k if k < F else 2 * F - 2 - k]
            if not GRAPH:
                fp.run(o, recs[i % 2], recs[(i + 1) % 2], ref_poc=i)
                continue
            key = (ci, k, i % 2)
            if key not in graphs:
                graphs[key] = ctx.record(lambda: fp.run(o, recs[i % 2], recs[(i + 1) % 2], ref_poc=k))
            ctx.replay(graphs[key])
    if GRAPH:
        for i in range(2 * (2 * F - 2)): step(i)
    for i in range(50): step(i)
    for c in ctxs: c.sync()
    N = 500
    t = time.perf_counter()
    for i in range(50, 50 + N): step(i)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t
    print("chains %d: %.1f frame passes/s (%.4f ms per picture)" % (K, K * N / dt, dt / (K * N) * 1e3))
    for c in ctxs: c.close()
