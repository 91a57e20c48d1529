#!/bin/bash
# several PROCESSES (own HIP runtimes) walking chains at once: is the launch path per process?
cd $GRAFT_REPO_ROOT
for np in 2 4 8; do
for k in 1 4; do
  pids=""
  for i in $(seq 1 $np); do
    python tools/cu_state_walk.py --mode chained --states 12000 --k $k --no-check > gpurun_out/l_walk_${np}_${k}_$i.json 2>/dev/null &
    pids="$pids $!"
  done
  wait $pids
  python - <<PY
import json,glob
tot=0; us=[]
for f in glob.glob("gpurun_out/l_walk_${np}_${k}_*.json"):
    d=json.load(open(f)); v=d["chains"]["$k"]; tot+=v["pictures_per_s"]; us.append(round(v["us_per_cu_state"],1))
print("processes $np x chains $k: sum pictures/s %.3f, us/state per process %s" % (tot, us))
PY
done
done
