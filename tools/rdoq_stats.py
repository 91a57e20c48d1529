#!/usr/bin/env python3
"""Workload statistics of the RDOQ stage on the bench content (run on the GPU
box): per transform block the number of levels and how far the coded region
reaches - what bounds the serial walk of quant_rdo_packed_kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd, qp = 1920, 1080, 10, int(os.environ.get("QP", "32"))
reps = int(os.environ.get("REPS", "3"))
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), 128)
fp = pipeline.FramePass(ctx, W, H, bd, qp=qp, rdoq=True)
for n in range(1, 1 + reps):
    O.upload(pad(clip.frame(n)), 128)
    fp.run(O, R, Rec, ref_poc=n - 1)
    ctx.sync()
    R, Rec = Rec, R
lv = fp.d_levels.to_array(np.int16, fp.n_levels)
d = fp.desc
off, _ = ctx.level_offsets(d.tx)
nnz = fp.d_nnz.to_array(np.int32, len(d.tx))
for comp, name in ((0, "luma 16x16"), (1, "chroma 8x8")):
    sel = [i for i in range(len(d.tx)) if d.tx[i]["comp"] == comp and d.tx[i]["w"] == (16 if comp == 0 else 8)]
    nz = nnz[sel]
    # live 4x4 sub-blocks (any level) and the highest anti-diagonal of the sub-block grid
    live, diag = [], []
    for i in sel[::7]:
        w, h = int(d.tx[i]["w"]), int(d.tx[i]["h"])
        b = lv[off[i]:off[i] + w * h].reshape(h, w) != 0
        sb = b.reshape(h // 4, 4, w // 4, 4).any(axis=(1, 3))
        live.append(int(sb.sum()))
        ys, xs = np.nonzero(sb)
        diag.append(int((ys + xs).max()) if len(ys) else -1)
    print("%s: %d blocks, cbf %.1f%%, mean nnz %.2f, p50/p90/p99/p99.9/max nnz %d/%d/%d/%d/%d; "
          "live sub-blocks mean %.2f; highest sub-block diagonal: %s" %
          (name, len(sel), 100.0 * (nz > 0).mean(), nz.mean(),
           *np.percentile(nz, [50, 90, 99, 99.9]), nz.max(),
           np.mean(live), np.bincount(np.array(diag) + 1)))
    print("   nnz histogram (0, 1-2, 3-5, 6-10, 11-20, 21-40, >40):",
          np.histogram(nz, [0, 1, 3, 6, 11, 21, 41, 100000])[0])
