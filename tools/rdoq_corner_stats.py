import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xvc_amd import api, pipeline, synth
W, H, bd, qp = 1920, 1080, 10, 32
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda pl: [np.ascontiguousarray(np.pad(p, 128 >> (1 if c else 0), mode="edge")) for c, p in enumerate(pl)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), 128)
fp = pipeline.FramePass(ctx, W, H, bd, qp=qp, rdoq=True)
for n in range(1, 4):
    O.upload(pad(clip.frame(n)), 128)
    fp.run(O, R, Rec, ref_poc=n - 1); ctx.sync(); R, Rec = Rec, R
cf = fp.d_coeffs.to_array(np.int16, fp.n_levels)
off, _ = ctx.level_offsets(fp.desc.tx)
fwd = [26214, 23302, 20560, 18396, 16384, 14564]
tot = live = corner = 0
for i, t in enumerate(fp.desc.tx):
    w, h, q = int(t["w"]), int(t["h"]), int(t["qp"])
    lw, lh = w.bit_length() - 1, h.bit_length() - 1
    qpb = q + 6 * (bd - 8)
    bias = (lw + lh) & 1
    sh = 14 + qpb // 6 + (15 - bd - ((lw + lh) >> 1)) + (7 if bias else 0)
    sc = fwd[qpb % 6] * (181 if bias else 1)
    a = np.abs(cf[off[i]:off[i] + w * h].astype(np.int64)).reshape(h, w)
    nz = ((a * sc + (1 << (sh - 1))) >> sh) != 0
    tot += 1
    if nz.any():
        live += 1
        out = nz.copy(); out[:4, :4] = False
        corner += (not out.any()) and w >= 8 and h >= 8
print("blocks %d live %d corner %d" % (tot, live, corner))
