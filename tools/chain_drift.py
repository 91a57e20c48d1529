import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd = 1920, 1080, 10
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
border = 128
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
F = 8
origs = []
for n in range(F):
    p = ctx.picture(W, H, bd); p.upload(pad(clip.frame(n)), border); origs.append(p)
a, b = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
a.upload(pad(clip.frame(0)), border)
fp = pipeline.FramePass(ctx, W, H, bd, qp=32, rdoq=True)
out = (C.c_int32 * 3)()
for j in range(600):
    k = j % (2 * F - 2); k = k if k < F else 2 * F - 2 - k
    fp.run(origs[k], a, b, ref_poc=j)
    a, b = b, a
    if j in (1, 5, 13, 27, 55, 111, 223, 447, 599):
        ctx.sync()
        ctx._check(ctx.lib.xvcgpu_quant_rdo_class_counts(ctx.h, out))
        ssd = fp.d_ssd.to_array(np.uint64, 2)
        print(j, "frame", k, list(out), "psnr %.2f" % pipeline.psnr_from_ssd(int(ssd[0]), int(ssd[1])))
