#!/usr/bin/env python3
"""What a kernel of the frame pass costs in THROUGHPUT: the same kernel from k
streams at once (k independent pictures), aggregate time per launch.  A kernel
bound by latency gets cheaper with k until some resource of the chip is full; the
value it settles at is its share of the pass when enough pictures are in flight.
(run on the GPU box)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = 1920, 1080, 10
K = int(os.environ.get("STREAMS", 6))
# CHAIN=<n>: every stream first codes n chained pictures (each against the previous
# reconstruction), so that the kernels are timed on the chain's steady state - what
# bench.py's default run spends its time on - instead of a chain's first picture
CHAIN = int(os.environ.get("CHAIN", 0))
ONLY = [k for k in os.environ.get("ONLY", "").split(",") if k]
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
ctxs = [api.Context(0) for _ in range(K)]
state = []
for i, c in enumerate(ctxs):
    O, R, Rec = (c.picture(W, H, bd) for _ in range(3))
    R.upload(pad(clip.frame(i % 7)), 128)
    O.upload(pad(clip.frame(i % 7 + 1)), 128)
    fp = pipeline.FramePass(c, W, H, bd, qp=32, rdoq=os.environ.get("QUANT", "rdoq") == "rdoq")
    for n in range(CHAIN):
        O.upload(pad(clip.frame((i + n) % 7 + 1)), 128)
        fp.run(O, R, Rec)
        c.sync()
        R, Rec = Rec, R
    O.upload(pad(clip.frame((i + CHAIN) % 7 + 1)), 128)
    fp.run(O, R, Rec)
    c.sync()
    state.append(dict(fp.kernel_steps(O, R, Rec)))
    if i == 0 and fp.rdoq:
        cc = (C.c_int32 * 3)()
        c._check(c.lib.xvcgpu_quant_rdo_class_counts(c.h, cc))
        print("RDOQ class lists of stream 0 (4 / 16 / 64 lanes per block):", list(cc))
names = [k for k in state[0].keys() if not ONLY or k in ONLY]
print("aggregate us per launch with k streams issuing the same kernel")
print("%-16s" % "kernel" + "".join("%8s" % ("k=%d" % k) for k in range(1, K + 1)))
total = [0.0] * K
for name in names:
    row = []
    for k in range(1, K + 1):
        n = 120
        for c in ctxs[:k]:
            c.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            for s in state[:k]:
                s[name]()
        for c in ctxs[:k]:
            c.sync()
        row.append((time.perf_counter() - t0) / (n * k) * 1e6)
        total[k - 1] += row[-1]
    print("%-16s" % name + "".join("%8.1f" % v for v in row))
print("%-16s" % "sum" + "".join("%8.1f" % v for v in total))
