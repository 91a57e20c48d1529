"""Why does bench.py's stream_decode see half of tools/dbg/dec_rate.py's rate?  The same
decode with bench's surroundings switched on one at a time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
what = sys.argv[1] if len(sys.argv) > 1 else "plain"
if "torch" in what:
    import torch
    torch.cuda.init()
    x = torch.zeros(16, device="cuda")
if "affinity" in what:
    print("affinity before:", len(os.sched_getaffinity(0)))
    os.sched_setaffinity(0, range(os.cpu_count()))
from xvc_amd import api, decoder
import stream_fixture as sf
ctx = api.Context(0)
extra = []
if "early" in what:         # the context's copy stream right behind its main stream
    ev = api.Event(ctx)
    ctx._check(ctx.lib.xvcgpu_upload_ahead(ctx.h, None, None, 0, None, ev.h))
if "ctxs" in what:
    for _ in range(3):
        c = api.Context(0); c.use_own_stream(); extra.append(c)
if "pics" in what:
    keep = [ctx.picture(1920, 1080, 10) for _ in range(40)]
fx = sf.StreamFixture("c1x")
w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
dec = decoder.PictureDecoder(ctx, w, h, bd)
infos = [fx.info[i] for i in range(fx.n)]
pos = {int(infos[i]["poc"]): i for i in range(fx.n)}
ri = np.full((fx.n, 2, 5), -1, np.int32)
pics = []
for i, info in enumerate(infos):
    ps, cs = sf.to_syntax(info, fx.cus(i))
    pics.append((ps, cs, np.ascontiguousarray(fx.levels(i))))
    for l in range(2):
        for k in range(int(info["num_ref"][l])):
            ri[i, l, k] = pos[int(info["ref_poc"][l][k])]
for k in range(2, 9):           # lanesK: K picture lanes
    if "lanes%d" % k in what:
        for _ in range(k - 1):
            c = api.Context(0); extra.append(c); dec.add_lane(c)
recs = [ctx.picture(w, h, bd) for _ in range(fx.n)]
def seq():
    dec.decode_sequence(pics, ri, recs); ctx.sync()
seq()
if "five" in what:          # bench's way: five repetitions, no warm-up beyond one
    t0 = time.perf_counter()
    for _ in range(5):
        seq()
    dt = (time.perf_counter() - t0) / 5
else:
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.0:
        seq(); n += 1
    dt = (time.perf_counter() - t0) / n
print(what, "%.1f pictures/s" % (fx.n / dt))
