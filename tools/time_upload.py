"""Host -> device cost of one 1080p input picture: (a) planes at the internal
depth through xvcgpu_picture_upload, (b) the application's packed 8-bit bytes
through xvcgpu_memcpy_h2d + xvcgpu_picture_import (+ xvcgpu_pad_border).
Pageable numpy memory on the host side (what a ctypes caller has)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xvc_amd import api, synth
W, H, bd = 1920, 1080, 10
ctx = api.Context(0)
clip = synth.SyntheticClip(W, H, bd)
planes = clip.frame(1)
P = ctx.picture(W, H, bd)
data8 = np.frombuffer(b"".join((p >> 2).astype(np.uint8).tobytes() for p in planes), np.uint8)
d_in = ctx.alloc(len(data8))
def a():
    P.upload(planes)
def b():
    ctx.h2d(d_in.ptr, data8)
    ctx._check(ctx.lib.xvcgpu_picture_import(ctx.h, P.h_pic, d_in.ptr, W, H, 8))
    ctx.pad_border(P)
    ctx.sync()
for name, fn, nbytes in (("upload 16-bit planes", a, W * H * 3), ("8-bit bytes + import + pad", b, W * H * 3 // 2)):
    for _ in range(5): fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(50): fn()
    ctx.sync()
    dt = (time.perf_counter() - t) / 50
    print("%-28s %.3f ms per picture  (%.1f GB/s over the link)" % (name, dt * 1e3, nbytes / dt / 1e9))
