"""How sparse are the blocks the RDOQ walk sees on the chain's steady state?
Per live 16x16 luma / 8x8 chroma block: the number of 4x4 sub-blocks holding a
coefficient that quantises to a non-zero value (the sub-blocks the walk has
decisions or costs to compute for), and the number of such coefficients.
Run on the GPU box."""
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
d = fp.desc
cf = fp.d_coeffs.to_array(np.int16, fp.n_levels)
off, _ = ctx.level_offsets(d.tx)
SCALES = [26214, 23302, 20560, 18396, 16384, 14564]
for comp, size in ((0, 16), (1, 8)):
    sel = np.flatnonzero((d.tx["comp"] == comp) & (d.tx["w"] == size) & (d.tx["h"] == size))
    nsb, nq = [], []
    for i in sel[::3]:
        qp = int(d.tx[i]["qp"]) + 6 * (bd - 8)
        lw = int(np.log2(size))
        shift = 14 + qp // 6 + (15 - bd - lw)
        a = np.abs(cf[off[i]:off[i] + size * size].astype(np.int64)).reshape(size, size)
        q = (a * SCALES[qp % 6] + (1 << (shift - 1))) >> shift
        sb = (q != 0).reshape(size // 4, 4, size // 4, 4).any(axis=(1, 3))
        nsb.append(int(sb.sum()))
        nq.append(int((q != 0).sum()))
    nsb, nq = np.array(nsb), np.array(nq)
    live = nsb > 0
    print("comp %d %dx%d: %d blocks sampled, live %.1f%%; sub-blocks with q != 0 per live block: "
          "hist %s; coefficients with q != 0 per live block: mean %.2f p50 %d p90 %d p99 %d max %d" %
          (comp, size, size, len(nsb), 100.0 * live.mean(),
           np.bincount(nsb[live], minlength=17).tolist(), nq[live].mean(),
           *np.percentile(nq[live], [50, 90, 99]), nq.max()))

# EvalLastPos' walk: from the last non-zero level back to the first level above 1 -
# how many sub-blocks does it pass (estimate by sub-block anti-diagonals)?
lv = fp.d_levels.to_array(np.int16, fp.n_levels)
for comp, size in ((0, 16), (1, 8)):
    sel = np.flatnonzero((d.tx["comp"] == comp) & (d.tx["w"] == size) & (d.tx["h"] == size))
    need, big, ones = [], [], []
    g = size // 4
    sy, sx = np.mgrid[0:g, 0:g]
    diag = (sx + sy)
    for i in sel[::3]:
        a = np.abs(lv[off[i]:off[i] + size * size].astype(np.int64)).reshape(size, size)
        if not a.any():
            continue
        nz = (a != 0).reshape(g, 4, g, 4).any(axis=(1, 3))
        gt = (a > 1).reshape(g, 4, g, 4).any(axis=(1, 3))
        d_last = diag[nz].max()
        d_stop = diag[gt].max() if gt.any() else 0
        need.append(int((nz & (diag >= d_stop) & (diag <= d_last)).sum()))
        big.append(int((a > 1).sum()))
        ones.append(int((a == 1).sum()))
    need = np.array(need)
    print("comp %d %dx%d: coded blocks %d; sub-blocks the EvalLastPos walk passes (coded ones between "
          "the highest level > 1 and the last level): hist %s mean %.2f; levels > 1 per block mean %.1f, "
          "levels == 1 mean %.1f" % (comp, size, size, len(need),
                                     np.bincount(need, minlength=g * g + 1).tolist(), need.mean(),
                                     np.mean(big), np.mean(ones)))
