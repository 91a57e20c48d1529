import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from xvc_amd import api, decoder
import stream_fixture as sf
ctx = api.Context(0)
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
recs = [ctx.picture(w, h, bd) for _ in range(fx.n)]
def seq():
    dec.decode_sequence(pics, ri, recs); ctx.sync()
def loop():
    for i, info in enumerate(infos):
        refs = [[recs[ri[i, l, k]] for k in range(int(info["num_ref"][l]))] for l in range(2)]
        dec.decode(pics[i][0], pics[i][1], pics[i][2], refs, recs[i])
    ctx.sync()
for name, f in (("sequence", seq), ("loop", loop)):
    f()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.0:
        f(); n += 1
    dt = (time.perf_counter() - t0) / n
    print(os.environ.get("XVC_DEC_TAIL_MIN_WAVES", "3"), name, "%.1f pictures/s" % (fx.n / dt))
