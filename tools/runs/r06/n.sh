#!/bin/bash
# round 6, run n: the engine's rates (and parity at k = 16)
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-n}
ENGINE_THREADS=${ENGINE_THREADS:-4} python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128 > gpurun_out/r06/${T}_walk_engine.json 2> gpurun_out/r06/${T}_walk_engine.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06/${T}_walk_engine.json"))
for k,v in d["chains"].items(): print(k, round(v["pictures_per_s"],3), round(v["launches_per_state"],2), v.get("matches_reference"), v.get("host_threads"), v.get("streams"))
PY
tail -2 gpurun_out/r06/${T}_walk_engine.err
