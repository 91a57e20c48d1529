#!/usr/bin/env python3
"""Per-picture timing of the C++ PictureDecoder on a stream fixture (run on the GPU
box): picture type, CUs, dependency waves, launches, host planning alone, and the
whole Decode (planning + upload + launches + device work) with a sync per picture."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stream_fixture as sf  # noqa: E402
from xvc_amd import api, decoder  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
fx = sf.StreamFixture(name)
ctx = api.Context(0)
w, h, bd = (int(fx.info[0][k]) for k in ("width", "height", "bitdepth"))
syn = [sf.to_syntax(fx.info[i], fx.cus(i)) for i in range(fx.n)]
lv = [fx.levels(i) for i in range(fx.n)]
dec = decoder.PictureDecoder(ctx, w, h, bd)
pics = [ctx.picture(w, h, bd) for _ in range(fx.n)]
TYPE = {0: "B", 1: "P", 2: "I"}
for rep in range(3):
    done, rows = {}, []
    for i in range(fx.n):
        info = fx.info[i]
        refs = [[done[int(info["ref_poc"][l][k])] for k in range(int(info["num_ref"][l]))]
                for l in range(2)]
        t0 = time.perf_counter()
        decoder.plan_picture(syn[i][0], syn[i][1], lv[i])
        t1 = time.perf_counter()
        dec.decode(syn[i][0], syn[i][1], lv[i], refs, pics[i])
        t2 = time.perf_counter()
        ctx.sync()
        t3 = time.perf_counter()
        done[int(info["poc"])] = pics[i]
        rows.append((i, int(info["poc"]), TYPE[int(syn[i][0]["pic_type"][0])], len(syn[i][1]),
                     dec.waves, dec.launches, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
                     1e3 * (t3 - t1)))
for r in rows:
    print("pic %2d poc %2d %s: %6d CUs %4d waves %5d launches  plan alone %.2f ms  "
          "Decode() returns after %.2f ms  done after %.2f ms" % r)
by = {}
for r in rows:
    by.setdefault(r[2], []).append(r[8])
print({k: "%d pictures, mean %.2f ms" % (len(v), sum(v) / len(v)) for k, v in by.items()},
      "overall %.1f pictures/s" % (len(rows) / (sum(r[8] for r in rows) * 1e-3)))
