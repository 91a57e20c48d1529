#!/bin/bash
# round 6, run by: affine search on 8-row slabs (a wave per 8 rows): parity, engine, chained walk, kernel stats
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q -k "affine or rd_calls or refs_forms or engine or cu_state" 2>&1 | tail -2
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['chains'].items(): print('engine k', k, round(v['pictures_per_s'],3), 'pictures/s', round(v['us_per_cu_state_aggregate'],2), 'us/state')"
python tools/cu_state_walk.py --mode chained --states 4000 --k 1 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['chains'].items(): print('chained k', k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,list))})" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
ENGINE_THREADS=4 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/by_engine -o walk -- python $R/tools/cu_state_walk.py --mode engine --states 1500 --k 16 --no-check > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/by_engine/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-50s calls %6s avg %7.1f us %5.1f %%" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
