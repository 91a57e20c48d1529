#!/usr/bin/env python3
"""Counts / timings of the RD replay per clip (GPU box): LIC bi steps and the
scratch-destination transform batch against the layered one."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import rd_fixture as rf  # noqa: E402
import rd_replay  # noqa: E402
import stream_fixture as sf  # noqa: E402
from test_gpu_me_calls import decode_stream  # noqa: E402
from xvc_amd import api  # noqa: E402

ctx = api.Context(0)
for name in sys.argv[1:] or ["tiny", "c0", "c1"]:
    fx = sf.StreamFixture(name)
    pics, w, h = decode_stream(ctx, fx)
    r = rd_replay.Replay(api, ctx, name, pics, w, h)
    r.bi_steps()
    r.timing = {}
    print(name, "bi steps (done, bad, skipped):", r.bi_steps(), {k: round(1e3 * v, 2) if isinstance(v, float) else v for k, v in r.timing.items()})
    r.timing = {}
    r.transform_calls_scratch(check=False)
    r.timing = {}
    t0 = time.time()
    print(name, "scratch (done, bad):", r.transform_calls_scratch(check=False),
          {k: round(1e3 * v, 2) if isinstance(v, float) else v for k, v in r.timing.items()},
          "wall %.2f s" % (time.time() - t0))
    r.destroy()
