import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rd_fixture as rf, rd_replay, stream_fixture as sf
from test_gpu_me_calls import decode_stream
from xvc_amd import api
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
ctx = api.Context(0)
fx = sf.StreamFixture(name)
pics, w, h = decode_stream(ctx, fx)
r = rd_replay.Replay(api, ctx, name, pics, w, h)
r.debug = collections.Counter()
r.debug_rows = []
print(r.transform_calls(max_layers=int(sys.argv[2]) if len(sys.argv) > 2 else None))
for k, v in sorted(r.debug.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
