#!/bin/bash
# round 6, run r: the default bench line (all legs)
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-r}
( time python bench.py > gpurun_out/r06/${T}_bench.json 2> gpurun_out/r06/${T}_bench.err ) 2> gpurun_out/r06/${T}_time.txt
tail -3 gpurun_out/r06/${T}_time.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r06/${T}_bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["cpu_baseline"])
print({k: d["roofline"][k] for k in ("kernel","achieved","frac","ms_per_launch")})
for k in ("stream_decode",):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if not isinstance(vv, (dict, list))})
rs = d.get("encoder_rd_serial") or (d.get("real_inputs") or {}).get("encoder_rd_serial")
def find(o, key):
    if isinstance(o, dict):
        if key in o: return o[key]
        for v in o.values():
            r = find(v, key)
            if r is not None: return r
    return None
rs = find(d, "encoder_rd_serial")
if rs:
    print("rd_serial us/state", rs.get("us_per_cu_state"), "engine", rs.get("engine", {}).get("pictures_per_s"))
    print("side_by_side", json.dumps(rs.get("side_by_side"))[:900])
    print("dag", rs.get("intra_picture_dag"))
    print("spent", rs.get("seconds_spent_measuring"))
sd = find(d, "stream_decode")
print("stream_decode", json.dumps(sd)[:600])
PY
tail -5 gpurun_out/r06/${T}_bench.err
