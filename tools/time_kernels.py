"""Isolated timings of the frame-pass kernels (env W, H, QP; default the 1080p
bench workload), with the algorithmic GB/s of the HBM-bound ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, pipeline, synth
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
QP, bd, border = int(os.environ.get("QP", 32)), 10, 128
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
pad = lambda planes: [np.ascontiguousarray(np.pad(p, border if c == 0 else border // 2, mode="edge")) for c, p in enumerate(planes)]
O, R, Rec = (ctx.picture(W, H, bd) for _ in range(3))
R.upload(pad(clip.frame(0)), border); O.upload(pad(clip.frame(1)), border)
fp = pipeline.FramePass(ctx, W, H, bd, qp=QP)
d = fp.desc
fp.run(O, R, Rec); ctx.sync()
def timed(fn, reps=50):
    fn(); ctx.sync(); ctx.timer_begin()
    for _ in range(reps): fn()
    return 1e3 * ctx.timer_end() / reps
which = sys.argv[1:] or ["me", "recon", "deblock", "pad", "ssd", "import8", "export8", "export8d", "crc", "variance", "histogram", "intra_satd", "intra_pred", "affine_me16", "affine_me32", "affine_me64"]
fns = {
    "me": lambda: ctx.me_search_dev(O, R, 3, fp.d_me.ptr, d.n_cus, fp.d_res.ptr, 16),
    "recon": lambda: ctx.recon_from_me_dev(O, R, Rec, fp.d_me.ptr, fp.d_res.ptr, d.n_cus, d.qp, d.qp_c, 0, fp.d_nnz.ptr, fp.d_cus_own),
    "deblock": lambda: ctx.deblock_dev(Rec, fp.d_cus.ptr, d.n_cus_total, fp.d_map.ptr, d.cu_map.shape[1]),
    "pad": lambda: ctx.pad_border(Rec),
    "ssd": lambda: ctx.picture_ssd_dev(O, Rec, 0, bd, fp.d_ssd.ptr),
}
Imp = ctx.picture(W, H, bd)   # import target: O stays the original for the kernels after it
d_in = ctx.alloc(W * H * 3)
d_out = ctx.alloc(W * H * 3)
d_small = ctx.alloc(8 * (W // 16 + 1) * (H // 16 + 1) * 2)
lib = ctx.lib
fns.update({
    "import8": lambda: lib.xvcgpu_picture_import(ctx.h, Imp.h_pic, d_in.ptr, W, H, 8),
    "export8": lambda: lib.xvcgpu_picture_export(ctx.h, Rec.h_pic, d_out.ptr, W, H, 8, 0),
    "export8d": lambda: lib.xvcgpu_picture_export(ctx.h, Rec.h_pic, d_out.ptr, W, H, 8, 1),
    "crc": lambda: lib.xvcgpu_picture_crc(ctx.h, Rec.h_pic, 0, d_small.ptr),
    "variance": lambda: lib.xvcgpu_variance_map(ctx.h, O.h_pic, d_small.ptr, 64, d_small.ptr + 8 * (W // 16 + 1) * (H // 16 + 1)),
    "histogram": lambda: lib.xvcgpu_histogram_distance(ctx.h, O.h_pic, R.h_pic, d_small.ptr),
})
ij = np.zeros((H // 16) * (W // 16), api.INTRA_DTYPE)
k = 0
for y in range(0, H - 15, 16):
    for x in range(0, W, 16):
        ij[k] = (x, y, 16, 16, 0, 34, (4 if x else 0) | (2 if y else 0) | (1 if x and y else 0),
                 16 if y and x + 32 <= W else 0, 0, 0)
        k += 1
d_ij = ctx.buffer(ij)
d_id = ctx.alloc(4 * 67 * len(ij))
fns.update({
    "intra_satd": lambda: lib.xvcgpu_intra_satd_batch(ctx.h, O.h_pic, R.h_pic, d_ij.ptr, len(ij), d_id.ptr, 16),
    "intra_pred": lambda: lib.xvcgpu_intra_pred_batch(ctx.h, R.h_pic, Rec.h_pic, d_ij.ptr, len(ij)),
})
# affine ME over the whole picture in CUs of one size, the predictor = the
# translational search result of the co-located 16x16 CU (as SearchRefIdx's
# mv_bootstrap does)
res = fp.d_res.to_array(api.MERES_DTYPE, d.n_cus)
def affine_jobs(cu):
    jobs = np.zeros((H // cu) * (W // cu), api.AFFINE_ME_DTYPE)
    k = 0
    for y in range(0, H - cu + 1, cu):
        for x in range(0, W - cu + 1, cu):
            b = jobs[k]
            b["x"], b["y"], b["w"], b["h"] = x, y, cu, cu
            b["lambda16"] = 200000
            if res is not None:
                r = res[(y // 16) * d.cus_per_row + x // 16]
                b["mvp"][:] = (int(r["mv_x"]), int(r["mv_y"]))
            k += 1
    return jobs[:k]
d_ar = ctx.alloc(api.AFFINE_ME_RESULT_DTYPE.itemsize * (H // 16) * (W // 16))
AO, AR = O, R
if os.environ.get("AFFINE_CONTENT") == "1":
    # zooming / rotating content (every CU iterates, sub-block MC) instead of the
    # translational clip
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_affine_me as oa
    o_, r_ = oa.warped_pics(np.random.default_rng(1), bd, W, H, border, 1.01, 0.004, (0.5, -0.25))
    chroma = np.full((H // 2 + border, W // 2 + border), 512, np.uint16)
    AO, AR = ctx.picture(W, H, bd), ctx.picture(W, H, bd)
    AO.upload([o_, chroma, chroma], border); AR.upload([r_, chroma, chroma], border)
    res = None
for cu in (16, 32, 64):
    aj = affine_jobs(cu)
    d_aj = ctx.buffer(aj)
    fns["affine_me%d" % cu] = (lambda d_aj=d_aj, n=len(aj): lib.xvcgpu_affine_me_batch(
        ctx.h, AO.h_pic, AR.h_pic, None, d_aj.ptr, n, d_ar.ptr))
N = W * H
alg = {"me": 4 * N, "recon": 9 * N + 16 * N // 16, "deblock": 6 * N + N,
       "pad": 2 * (2 * 128 * (W + H + 256) + 4 * 64 * (W // 2 + H // 2 + 128)), "ssd": 4 * N,
       "import8": 1.5 * N * 3, "export8": 1.5 * N * 3, "export8d": 1.5 * N * 3, "crc": 3 * N,
       "variance": 2 * N, "histogram": 4 * N, "intra_satd": 4 * N, "intra_pred": 4 * N,
       "affine_me16": 4 * N, "affine_me32": 4 * N, "affine_me64": 4 * N}
for k in which:
    us = timed(fns[k])
    print("%-8s %8.2f us  %7.0f GB/s algorithmic" % (k, us, alg[k] / us / 1e3))
