import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from xvc_amd import api
import rd_serial, stream_fixture as sf
from test_gpu_me_calls import decode_stream
name, poc = sys.argv[1], int(sys.argv[2]); nst = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
ctx = api.Context(0)
fx = sf.StreamFixture(name)
pics, w, h = decode_stream(ctx, fx)
sp = rd_serial.SerialPicture(api, name, poc)
run = rd_serial.ChainedRun(api, ctx, sp, pics, w, h, rd_serial.ref_lists_of(name, poc))
n = min(nst, len(sp.states))
run.run_chained(0, n, False)
R = run.cres["results"][0]; st = sp.states; cd_all = sp.tabs["cands"]
cnt = collections.Counter(); shown = 0
for ns in range(n):
    s = st[ns]
    if not s["supported"] or s["kind"] < 2 or sp.pass_count[ns] == 0: continue
    pf = int(sp.pass_first[ns])
    cds = cd_all[int(s["cand_first"]):int(s["cand_first"]) + int(s["cand_count"])]
    for c in cds:
        pi = pf + (1 if c["kind"] >= 2 else 0); r = R[pi]; l, k = int(c["list"]), int(c["ref_idx"])
        nc = 3 if c["kind"] >= 2 else 1
        if c["kind"] in (0, 2):
            f = dict(dist=r["dist"][l, k] == c["dist"], bits=r["bits"][l, k] == c["bits"], mvp=r["mvp_idx"][l, k] == c["mvp_idx"],
                     start=r["start_idx"][l, k] == c["start_mvp_idx"], mv=np.array_equal(r["mv"][l, k][:nc], c["mv"][:nc]))
        else:
            f = dict(slist=r["search_list"] == l, dist=r["bi_dist"][k] == c["dist"], bits=r["bi_bits"][k] == c["bits"],
                     mvp=r["bi_mvp_idx"][k] == c["mvp_idx"], mv=np.array_equal(r["bi_mv"][k][:nc], c["mv"][:nc]))
        badf = tuple(sorted(a for a, v in f.items() if not v))
        cnt[(int(c["kind"]), int(c["reused"]), int(c["flags"]), badf)] += 1
        if badf and shown < 6:
            shown += 1
            print("state", ns, tuple(s)[:8], "cand kind", c["kind"], "l", l, "r", k, "reused", c["reused"], "flags", c["flags"], badf)
            print("  want dist", c["dist"], "bits", c["bits"], "mvp_idx", c["mvp_idx"], "start", c["start_mvp_idx"], "mv", c["mv"][:nc].tolist(), "mvp", c["mvp"][:, :nc].tolist())
            if c["kind"] in (0, 2):
                print("  got  dist", r["dist"][l, k], "bits", r["bits"][l, k], "mvp_idx", r["mvp_idx"][l, k], "start", r["start_idx"][l, k], "mv", r["mv"][l, k][:nc].tolist())
            else:
                print("  got  slist", r["search_list"], "dist", r["bi_dist"][k], "bits", r["bi_bits"][k], "mvp_idx", r["bi_mvp_idx"][k], "mv", r["bi_mv"][k][:nc].tolist(), "best_ref", r["best_ref"], "cost_list", r["cost_list"])
            p = sp.passes[pi]
            print("  pass", pi, "flags", p["flags"], "uni_job", p["uni_job"].tolist(), "start_dist", p["start_dist"].tolist(), "bi_job", p["bi_job"], "slot", p["slot"].tolist(), "same", p["same_poc_in_l0"].tolist())
            if c["kind"] == 0:
                sdv = run.ctx.alloc(8)  # dummy
                buf = api.DeviceBuffer.__new__(api.DeviceBuffer); buf.ctx = ctx; buf.ptr = run.d["start_dist"]; buf.nbytes = 8 * sp.n_start_dist
                allsd = buf.to_array(np.uint64, sp.n_start_dist)
                i0 = int(p["start_dist"][l, k])
                cands = sp.start_cands[i0:i0 + 2]
                direct = ctx.mc_metric_batch(run.orig, run.refs[int(p["slot"][l, k])], cands)
                print("  start_dist dev", allsd[i0:i0 + 2].tolist(), "direct", direct.tolist(), "cands", cands.tolist())
for k, v in sorted(cnt.items()): print(k, v)
