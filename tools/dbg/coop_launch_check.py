import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
N_OTHERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
import numpy as np
import stream_fixture as sf
from xvc_amd import api, decoder
fx = sf.StreamFixture("c1x")
ctx = api.Context(0)
others = [api.Context(0) for _ in range(N_OTHERS)]
w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
dec = decoder.PictureDecoder(ctx, w, h, bd)
ps, cs = sf.to_syntax(fx.info[0], fx.cus(0))
rec = ctx.picture(w, h, bd)
LV = fx.levels(0)
for i in range(3):
    t0 = time.perf_counter()
    dec.decode(ps, cs, LV, [[], []], rec)
    ctx.sync()
    print("other contexts", N_OTHERS, "launches", dec.launches, "ms %.2f" % (1e3 * (time.perf_counter() - t0)))
