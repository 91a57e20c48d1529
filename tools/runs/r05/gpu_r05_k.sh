#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 900 python tools/cu_state_walk.py --mode chained --states 4000 --k 1,4 --no-check > gpurun_out/k_walk.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/k_walk.json"))
print("HIP_FORCE_DEV_KERNARG=$v walk:", {k:(round(v["pictures_per_s"],3), round(v["us_per_cu_state"],1)) for k,v in d["chains"].items()})
PY
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 300 --warmup 30 --no-decode --no-cpu > gpurun_out/k_bench.json 2>/dev/null
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/k_bench.json") if x.startswith("{")][-1])
print("HIP_FORCE_DEV_KERNARG=$v bench:", round(d["value"],1))
PY
done
