"""Launches the RDOQ step of the frame pass a few times on the chain's steady
state (CHAIN pictures coded against each other first), for rocprofv3 --pmc
(tools/pmc_rdoq.sh).  Run on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = 1920, 1080, 10
CHAIN = int(os.environ.get("CHAIN", 120))
QP = int(os.environ.get("QP", 32))
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), 128)
fp = pipeline.FramePass(ctx, W, H, bd, qp=QP, rdoq=True)
for n in range(CHAIN + 1):
    O.upload(pad(clip.frame(n % 7 + 1)), 128)
    fp.run(O, R, Rec)
    ctx.sync()
    if n < CHAIN:
        R, Rec = Rec, R
steps = dict(fp.kernel_steps(O, R, Rec))
for _ in range(int(os.environ.get("REPS", 5))):
    steps[os.environ.get("STEP", "quant_rdo")]()
ctx.sync()
