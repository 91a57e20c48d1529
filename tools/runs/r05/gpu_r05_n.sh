#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/n_prof -o eng -- python $R/tools/cu_state_walk.py --mode engine --states 2000 --k 48 --no-check > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/n_prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f ms" % (tot/1e6))
    for r in rows[:16]:
        print("%-52s %7s %8.1f %5.1f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
for f in glob.glob("$R/gpurun_out/n_prof/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows=[r for r in rows if "cs_seg" in r["Kernel_Name"] or "memcpy" in r["Kernel_Name"].lower()]
    t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
    print("span of the seg kernels %.1f ms, their summed duration %.1f ms (busy %.0f %%), launches %d" % ((t1-t0)/1e6, busy/1e6, 100*busy/(t1-t0), len(rows)))
PY
rm -rf $R/gpurun_out/n_prof
