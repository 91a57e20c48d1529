#!/usr/bin/env python3
"""Two chains of frame passes, one with the all-zero proof off and one with it on:
the first picture and block whose levels differ (there must be none).  (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xvc_amd import api, pipeline, synth  # noqa: E402

W, H, bd = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 10
QP = int(os.environ.get("QP", 32))
N = int(os.environ.get("N", 300))
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge"))
                  for c, p in enumerate(pl)]
ctxs = [api.Context(0), api.Context(0)]
ctxs[0].set_rdoq_prove_zero(0)
ctxs[1].set_rdoq_prove_zero(int(os.environ.get("MODE", -1)))
fps, pics = [], []
for c in ctxs:
    fps.append(pipeline.FramePass(c, W, H, bd, qp=QP, rdoq=True))
    O, R, Rec = (c.picture(W, H, bd) for _ in range(3))
    R.upload(pad(clip.frame(0)), 128)
    pics.append([O, R, Rec])
for n in range(1, N + 1):
    k = n % 14
    f = pad(clip.frame(k if k < 8 else 14 - k))
    lv = []
    for c, fp, p in zip(ctxs, fps, pics):
        p[0].upload(f, 128)
        fp.run(p[0], p[1], p[2], ref_poc=n - 1)
        c.sync()
        lv.append((fp.d_levels.to_array(np.int16, fp.n_levels),
                   fp.d_nnz.to_array(np.int32, len(fp.desc.tx)),
                   fp.d_coeffs.to_array(np.int16, fp.n_levels)))
        p[1], p[2] = p[2], p[1]
    if not np.array_equal(lv[0][0], lv[1][0]) or not np.array_equal(lv[0][1], lv[1][1]):
        off = fps[0].d_level_off.to_array(np.uint32, len(fps[0].desc.tx)).astype(np.int64)
        bad = np.flatnonzero(lv[0][1] != lv[1][1])
        print("picture", n, "blocks with other counts:", len(bad), bad[:10])
        b = int(bad[0])
        tx = fps[0].desc.tx[b]
        w, h = int(tx["w"]), int(tx["h"])
        print("block", b, tx)
        print("coefficients (proof off run):\n", lv[0][2][off[b]:off[b] + w * h].reshape(h, w))
        print("levels off:\n", lv[0][0][off[b]:off[b] + w * h].reshape(h, w))
        print("levels on:\n", lv[1][0][off[b]:off[b] + w * h].reshape(h, w))
        np.savez(os.path.join(ROOT, "gpurun_out", "prove_diff.npz"),
                 coeffs=lv[0][2][off[b]:off[b] + w * h].reshape(h, w), tx=tx,
                 prm=fps[0].desc.rdoq_params[b], ctx=fps[0].desc.rdoq_contexts)
        sys.exit(1)
print("no difference over", N, "pictures")
