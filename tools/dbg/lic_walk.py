#!/usr/bin/env python3
"""The walk over tiny's states (LIC states included) in the serial, chained, by-state and
live forms: counts per table, and where a LIC state differs from the capture."""
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

name, poc = sys.argv[1] if len(sys.argv) > 1 else "tiny", 2
ctx = api.Context(0)
pics, w, h = decode_stream(ctx, sf.StreamFixture(name))
sp = rd_serial.SerialPicture(api, name, poc)
print(sp.summary(), "LIC states", int(((sp.states["flags"] & 2) != 0).sum()))
run = rd_serial.SerialRun(api, ctx, sp, pics, w, h)
stats = run.run_serial(0, len(sp.states))
res = run.check(0, len(sp.states))
print("serial", res, "%.1f us / state" % (1e6 * stats.seconds / stats.states), getattr(run, "first_bad_intra", None))
print("by kind us", [round(1e6 * stats.seconds_by_kind[i] / max(stats.states_by_kind[i], 1), 1) for i in range(5)],
      [stats.states_by_kind[i] for i in range(5)])
if res.get("intra_calls", (0, 0))[1]:
    w = sp.in_want
    g_n, g_d = run.res["in_nnz"], run.res["in_dist"]
    bad = (g_n != w["nnz"]) | ((w["completed"] != 0) & (g_d != w["dist"]))
    print("intra calls bad by comp", [int((bad & (w["comp"] == c)).sum()) for c in range(3)],
          "LM", int((bad & (w["mode"] == 67)).sum()), "of", int((w["mode"] == 67).sum()),
          "nnz-only", int((g_n != w["nnz"]).sum()))
    for k in np.flatnonzero(bad)[:6]:
        print("  itx", k, tuple(w[k][["x", "y", "w", "h", "comp", "mode", "tx_skip", "tx_hor", "tx_ver", "nnz", "dist"]]), int(g_n[k]), int(g_d[k]))
st = sp.states
lic = (st["flags"] & 2) != 0
# bi mismatches by state
g, wnt = run.res["bi_res"], sp.bi_want
bad = (g["mv_x"] != wnt["mv"][:, 0, 0]) | (g["mv_y"] != wnt["mv"][:, 0, 1]) | (g["subpel_dist"] != wnt["dist"])
print("bi bad", int(bad.sum()), "of", len(bad), "in LIC steps", int((bad & ((wnt["flags"] & 2) != 0)).sum()))
for k in np.flatnonzero(bad)[:5]:
    print("  bi", k, tuple(wnt[k][["x", "y", "w", "h", "flags", "nb_index"]]), tuple(g[k]), wnt[k]["mv"][0], wnt[k]["dist"])
dz_w = sp.ev_want["dist_zero"]
dz_g = run.res["ev_dz_dist"].reshape(-1, 3)
valid = dz_w != np.uint64(0xffffffffffffffff)
badz = ((dz_g != dz_w) & valid).any(1)
evlic = (sp.ev_want["flags"] & 2) != 0
print("dist_zero bad", int(badz.sum()), "LIC evals", int(evlic.sum()), "bad among LIC", int((badz & evlic).sum()))
for k in np.flatnonzero(badz)[:5]:
    print("  ev", k, tuple(sp.ev_want[k][["x", "y", "w", "h", "inter_dir", "flags", "nb_index"]]), dz_g[k], dz_w[k])
run.destroy()
lists = rd_serial.ref_lists_of(name, poc)
for label, kw in (("by position", dict(by_position=True)), ("by state", dict(by_position=False)),
                  ("live", dict(by_position=False, live=True))):
    run = rd_serial.ChainedRun(api, ctx, sp, pics, w, h, lists)
    stats = run.run_chained(0, len(sp.states), **kw)
    res = run.check(0, len(sp.states), searches=False)
    res.update(run.check_chained(0, len(sp.states)))
    print(label, res, "%.1f us / state, %.2f round trips per state" % (
        1e6 * stats.seconds / stats.states, stats.round_trips / stats.states), getattr(run, "first_bad", None),
        getattr(run, "first_bad_intra", None))
    run.destroy()
