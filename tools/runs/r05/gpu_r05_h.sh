#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cu_state.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25
for m in chained live; do
timeout 900 python tools/cu_state_walk.py --mode $m --states 4000 --k 1,4 --no-check > gpurun_out/h_walk_$m.json 2> gpurun_out/h_walk_$m.err
python - <<PY
import json
d=json.load(open("gpurun_out/h_walk_$m.json"))
print("$m:", {k:(round(v["pictures_per_s"],3), round(v["us_per_cu_state"],1), round(v["round_trips_per_state"],2), round(v["api_calls_per_state"],2)) for k,v in d["chains"].items()})
PY
done
