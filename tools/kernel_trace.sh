#!/bin/bash
# Per-kernel durations of the frame pass with ONE picture in flight (no overlap
# between chains: a kernel's duration here is its own latency).  Run on the GPU
# box: tools/kernel_trace.sh [bench args] ; prints the rocprofv3 stats table.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ktrace
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python $R/bench.py --no-cpu --no-decode --chains 1 --steps 60 --warmup 10 "$@" > $out/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %5s avg %9.1f us min %8.1f max %8.1f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
