"""Chained 1080p frame passes on the GPU against the oracle (developer check)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, oracle_frame
from xvc_amd import api, pipeline, synth
W, H, bd, BL = 1920, 1080, 10, 128
nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pad = lambda planes: [np.ascontiguousarray(np.pad(p, BL if c == 0 else BL // 2, mode="edge")) for c, p in enumerate(planes)]
ctx = api.Context(0); xo = ol.Lib("xo")
clip = synth.SyntheticClip(W, H, bd)
fp = pipeline.FramePass(ctx, W, H, bd, qp=32)
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
ref_host = pad(clip.frame(0)); R.upload(ref_host, BL)
for n in range(1, nframes + 1):
    orig_host = pad(clip.frame(n)); O.upload(orig_host, BL)
    fp.run(O, R, Rec, ref_poc=n - 1); ctx.sync()
    res, nnz, cus, ssd = fp.results()
    e_rec, e_res, e_nnz, e_cus, e_ssd = oracle_frame.frame_pass(fp.desc, bd, orig_host, ref_host, BL, ref_poc=n - 1, lib=xo)
    bad = np.nonzero(res != e_res)[0]
    got = Rec.download(BL)
    print("frame", n, "me mismatches", len(bad), "rec equal", all(np.array_equal(got[c], e_rec[c]) for c in range(3)),
          "ssd", tuple(int(v) for v in ssd), e_ssd, "psnr %.3f" % pipeline.psnr_from_ssd(*e_ssd))
    if len(bad):
        for i in bad[:5]:
            print("  cu", i, fp.desc.me[i], res[i], e_res[i])
    ref_host = got  # continue from the GPU's own reconstruction
    R, Rec = Rec, R
