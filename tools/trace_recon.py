"""Per-job phase timeline of recon_from_me (developer build, -DXVCGPU_TRACE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H, bd, border = 1920, 1080, 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd)
d = fp.desc
lib = api.load_library()
ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16)
run = lambda: ctx.recon_from_me_dev(O, R, Rec, fp.d_me.ptr, fp.d_res.ptr, d.n_cus, d.qp, d.qp_c, 0, fp.d_nnz.ptr, fp.d_cus_own)
for _ in range(3): run()
ctx.sync(); ctx.timer_begin(); run(); ms = ctx.timer_end()
n = 3 * d.n_cus
buf = np.zeros((n, 16), np.uint64)
lib.xvcgpu_debug_me_trace(buf.ctypes.data_as(C.c_void_p), n)
t = buf[:, :9].astype(np.int64)
names = ["desc+interp", "residual", "fwd 1", "fwd 2", "quant", "dequant", "inverse", "addclip+store"]
for comp, nm in ((0, "luma"), (1, "chroma U")):
    tt = t[comp::3]
    ok = (tt[:, 1:] > 0).all(axis=1)   # jobs that ran every phase (nnz != 0, not dc-only)
    tt = tt[ok]
    life = tt[:, 8] - tt[:, 0]
    print("%s: kernel %.4f ms, %d full jobs, lifetime mean %.0f ticks" % (nm, ms, len(tt), life.mean()))
    for k, pn in enumerate(names):
        dph = tt[:, k + 1] - tt[:, k]
        print("  %-14s mean %7.0f  p50 %7.0f  share %5.1f%%" % (pn, dph.mean(), np.median(dph), 100.0 * dph.sum() / life.sum()))
