#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CHAIN=15 REPS=20 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/e_prof -o rq -- python $R/tools/run_rdoq_steady.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/e_prof/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # the last 20 repetitions of the step: find the last 20 quant_rdo_packed4 launches and the kernels around them
    idx=[i for i,r in enumerate(rows) if "quant_rdo_packed4" in r["Kernel_Name"]][-20:]
    first=idx[0]-2
    seg=rows[first:]
    acc=collections.defaultdict(list)
    prev_end=None
    gaps=[]
    for r in seg:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        acc[r["Kernel_Name"].split("(")[0][:40]].append((e-s)/1e3)
        if prev_end is not None: gaps.append((s-prev_end)/1e3)
        prev_end=e
    for k,v in acc.items(): print("%-42s n=%d mean %.1f us" % (k,len(v),sum(v)/len(v)))
    print("gaps between consecutive kernels: mean %.1f us (n=%d)" % (sum(gaps)/len(gaps), len(gaps)))
    span=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3
    print("span per step: %.1f us" % (span/20))
PY
