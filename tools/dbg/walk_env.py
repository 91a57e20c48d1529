"""tools/cu_state_walk.py's chained k = 4 figure with bench.py's surroundings switched on."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
what = sys.argv[1] if len(sys.argv) > 1 else "plain"
if "torch" in what:
    import torch
    torch.cuda.init(); x = torch.zeros(4, device="cuda")
from xvc_amd import api
import cu_state_walk
extra = []
if "ctxs" in what:
    for _ in range(5):
        c = api.Context(0); c.use_own_stream(); extra.append(c)
if "null" in what:
    c = api.Context(0); extra.append(c)
    p = c.picture(1920, 1080, 10)
if "serial" in what:
    cu_state_walk.walk(api, "c1", 2, 2000, [1, 4], "serial", check=False)
r = cu_state_walk.walk(api, "c1", 2, 4000, [4], "chained", check=False)
print(what, round(r["chains"]["4"]["pictures_per_s"], 3), round(r["chains"]["4"]["one_thread"]["pictures_per_s"], 3))
