#!/usr/bin/env python3
"""Is the GPU full while three RDOQ picture chains are in flight?  Runs the
three chains (as bench.py does) with and without a fourth stream that only
repeats one throughput-bound kernel (me_search), and reports what each got.
(run on the GPU box)"""
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
NCH = int(os.environ.get("CHAINS", "3"))
chains = []
F = 8
ctx0 = api.Context(0)
origs = []
for n in range(1, F + 1):
    p = ctx0.picture(W, H, bd)
    p.upload(pad(clip.frame(n)), 128)
    origs.append(p)
CYC = 2 * F - 2


def orig_at(j):
    k = j % CYC
    return origs[k if k < F else CYC - k]


for k in range(NCH + 1):
    c = api.Context(0)
    O = origs[0]
    recs = [c.picture(W, H, bd), c.picture(W, H, bd)]
    phase = (k * CYC) // NCH
    recs[0].upload(pad(clip.frame(phase if phase < F else CYC - phase)), 128)
    fp = pipeline.FramePass(c, W, H, bd, qp=32, rdoq=os.environ.get("QUANT", "rdoq") == "rdoq")
    chains.append((c, (k * CYC) // NCH, recs, fp))


def run(steps, filler):
    fc, _, frecs, ffp = chains[NCH]
    frecs[1].upload(pad(clip.frame(3)), 128)
    fill = dict(ffp.kernel_steps(origs[2], frecs[1], frecs[0]))["me_search"]
    for c in chains:
        c[0].sync()
    t0 = time.perf_counter()
    nfill = 0
    for i in range(steps):
        c, phase, recs, fp = chains[i % NCH]
        j = i // NCH
        fp.run(orig_at(j + phase), recs[j % 2], recs[(j + 1) % 2], ref_poc=j)
        if filler and i % filler == 0:
            fill()
            nfill += 1
    for c in chains:
        c[0].sync()
    dt = time.perf_counter() - t0
    return steps / dt, nfill / dt


run(150, 0)
a, _ = run(900, 0)
print("chains alone: %.0f passes/s" % a)
for f in (3, 2, 1):
    b, m = run(900, f)
    print("with a me_search every %d steps on a 4th stream: %.0f passes/s + %.0f searches/s "
          "(a search alone: 80 us)" % (f, b, m))
