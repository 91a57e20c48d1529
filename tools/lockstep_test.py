#!/usr/bin/env python3
"""Does the ORDER in which the host issues the kernels of k picture chains matter?
  pass-major: all kernels of chain 0's pass, then chain 1's, ... (what bench.py does)
  kind-major: the motion searches of all chains, then their forward transforms, ...
Kernels of the same kind run well beside each other (tools/throughput_cost.py);
kind-major issue keeps the chains in step.  (run on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = 1920, 1080, 10
K = int(os.environ.get("CHAINS", 3))
N = int(os.environ.get("STEPS", 300))
rdoq = os.environ.get("QUANT", "rdoq") == "rdoq"
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
ctxs = [api.Context(0) for _ in range(K)]
F = 8
chains = []
for i, c in enumerate(ctxs):
    origs = []
    for n in range(F):
        p = c.picture(W, H, bd)
        p.upload(pad(clip.frame(n + 1)), 128)
        origs.append(p)
    a, b = c.picture(W, H, bd), c.picture(W, H, bd)
    a.upload(pad(clip.frame(0)), 128)
    fp = pipeline.FramePass(c, W, H, bd, qp=32, rdoq=rdoq)
    # the launches of a pass for every (original, ping-pong parity)
    steps = {}
    for k in range(F):
        steps[(k, 0)] = fp.kernel_steps(origs[k], a, b)
        steps[(k, 1)] = fp.kernel_steps(origs[k], b, a)
    chains.append((c, fp, steps, i * 2))


def frame_of(j, phase):
    k = (j + phase) % (2 * F - 2)
    return k if k < F else 2 * F - 2 - k


events = [[api.Event(c) for _ in range(2)] for c, *_ in chains]


def run(kind_major, n, barrier=False):
    for c, *_ in chains:
        c.sync()
    t0 = time.perf_counter()
    for j in range(n):
        lists = [st[(frame_of(j, ph), j & 1)] for _, _, st, ph in chains]
        if barrier and j:
            # every chain starts its pass when all chains have finished the previous one
            for i, (c, *_) in enumerate(chains):
                for o in range(K):
                    if o != i:
                        events[o][(j - 1) & 1].wait(c)
        if barrier:
            for s in range(len(lists[0])):
                for l in lists:
                    l[s][1]()
            for i, (c, *_) in enumerate(chains):
                events[i][j & 1].record(c)
        elif kind_major:
            for s in range(len(lists[0])):
                for l in lists:
                    l[s][1]()
        else:
            for l in lists:
                for _, fn in l:
                    fn()
    for c, *_ in chains:
        c.sync()
    return n * K / (time.perf_counter() - t0)


run(False, 20)
for _ in range(2):
    print("pass-major %.0f passes/s   kind-major %.0f passes/s   kind-major + barrier per pass %.0f passes/s" %
          (run(False, N), run(True, N), run(True, N, True)))
