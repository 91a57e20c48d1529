import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd = 1920, 1080, 10
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
border = 128
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
fp = pipeline.FramePass(ctx, W, H, bd, qp=32, rdoq=True)
for i in range(4):
    R.upload(pad(clip.frame(i)), border); O.upload(pad(clip.frame(i + 1)), border)
    fp.run(O, R, Rec); ctx.sync()
    out = (C.c_int32 * 3)()
    ctx._check(ctx.lib.xvcgpu_quant_rdo_class_counts(ctx.h, out))
    print(i, list(out), "of", len(fp.desc.tx))
