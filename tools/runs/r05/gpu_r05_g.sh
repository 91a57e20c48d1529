#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 8 16 32; do
  GPU_MAX_HW_QUEUES=$q timeout 900 python tools/cu_state_walk.py --mode chained --states 4000 --k 1,4,8,16 --no-check > gpurun_out/g_walk_q$q.json 2> gpurun_out/g_walk_q$q.err
  python - <<PY
import json
d=json.load(open("gpurun_out/g_walk_q$q.json"))
print("queues $q:", {k:(round(v["pictures_per_s"],3), round(v.get("us_per_state",0),1)) for k,v in d["chains"].items()})
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g_prof -o walk -- python $R/tools/cu_state_walk.py --mode chained --states 3200 --k 1 --no-check > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/g_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("kernel, calls, avg us, % of kernel time")
    for r in rows[:16]:
        print("%-60s %7s %8.1f %5.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
