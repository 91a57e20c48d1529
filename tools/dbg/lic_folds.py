#!/usr/bin/env python3
"""Which fields of a LIC state's priced candidates differ from the capture (chained form,
XVC_CS_LIC passes through the device folds)."""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rd_serial
import stream_fixture as sf
from test_gpu_me_calls import decode_stream
from xvc_amd import api

name, poc = "tiny", 2
ctx = api.Context(0)
pics, w, h = decode_stream(ctx, sf.StreamFixture(name))
sp = rd_serial.SerialPicture(api, name, poc)
run = rd_serial.ChainedRun(api, ctx, sp, pics, w, h, rd_serial.ref_lists_of(name, poc))
stats = run.run_chained(0, len(sp.states), by_position=False)
res = run.check(0, len(sp.states), searches=False)
res.update(run.check_chained(0, len(sp.states)))
print({k: v for k, v in res.items()})
st = sp.states
R = run.cres["results"][0]
cd_all = sp.tabs["cands"]
cnt = {}
shown = 0
for ns in range(len(st)):
    s = st[ns]
    if not s["supported"] or s["kind"] not in (2, 3) or sp.pass_count[ns] == 0:
        continue
    lic = bool(int(s["flags"]) & 2)
    pf = int(sp.pass_first[ns])
    cds = cd_all[int(s["cand_first"]):int(s["cand_first"]) + int(s["cand_count"])]
    for c in cds:
        pi = pf + (1 if c["kind"] >= 2 else 0)
        r, l, k = R[pi], int(c["list"]), int(c["ref_idx"])
        if c["kind"] in (0, 2):
            f = {"dist": r["dist"][l, k] == c["dist"], "bits": r["bits"][l, k] == c["bits"],
                 "mvp_idx": r["mvp_idx"][l, k] == c["mvp_idx"], "start": r["start_idx"][l, k] == c["start_mvp_idx"],
                 "mv": np.array_equal(r["mv"][l, k][:1], c["mv"][:1])}
        else:
            f = {"search_list": r["search_list"] == l, "dist": r["bi_dist"][k] == c["dist"], "bits": r["bi_bits"][k] == c["bits"],
                 "mvp_idx": r["bi_mvp_idx"][k] == c["mvp_idx"], "mv": np.array_equal(r["bi_mv"][k][:1], c["mv"][:1])}
        for name_, ok in f.items():
            key = (lic, int(c["kind"]), name_)
            a = cnt.setdefault(key, [0, 0])
            a[0] += 1
            a[1] += 0 if ok else 1
        if lic and not all(f.values()) and shown < 6:
            shown += 1
            print("state", ns, tuple(s[["x", "y", "w", "h", "flags", "kind"]]), "cand kind", int(c["kind"]), "l", l, "r", k,
                  {k_: bool(v) for k_, v in f.items()}, "want dist/bits/mvp/start", int(c["dist"]), int(c["bits"]),
                  int(c["mvp_idx"]), int(c["start_mvp_idx"]), "mv", c["mv"][0].tolist(), "reused", int(c["reused"]),
                  "got", int(r["dist"][l, k]) if c["kind"] == 0 else int(r["bi_dist"][k]),
                  int(r["bits"][l, k]) if c["kind"] == 0 else int(r["bi_bits"][k]),
                  int(r["start_idx"][l, k]), r["mv"][l, k][0].tolist() if c["kind"] == 0 else r["bi_mv"][k][0].tolist())
for key in sorted(cnt):
    if cnt[key][1]:
        print(key, cnt[key])
sd = run.cres["start_dist"][0]
print("start_dist sample", sd[:12].tolist())
