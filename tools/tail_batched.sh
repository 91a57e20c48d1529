#!/bin/bash
# tools/tail_batched.sh <out.txt>: the batched tail's HBM fraction at 1080p and 4320p
# (tools/tail_batched.py under rocprofv3 --kernel-trace --stats; run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$R/gpurun_out/tail_batched.txt}
: > $out
cd /tmp && export TMPDIR=/tmp
for cfg in "4 6 1920 1080 10" "4 2 7680 4320 5"; do
  set -- $cfg
  d=/tmp/tb_$3; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o tb -- python $R/tools/tail_batched.py $cfg > $d.log 2>&1
  python - >> $out <<PY
import csv, glob, re
line = [l for l in open("$d.log") if l.startswith("TAIL_BATCHED")][0]
m = dict(kv.split("=") for kv in line.split()[1:] if "=" in kv)
B, alg = int(m["B"]), float(m["alg_bytes_per_picture"])
f = glob.glob("$d/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "deblock_tail_multi_kernel" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        gbs = B * alg / (us * 1e-6) / 1e9
        print("deblock_tail_multi_kernel, $3x$4, %d pictures per launch, %s groups of distinct pictures in turn: "
              "%s launches, %.1f us per launch (rocprofv3 kernel stats), %.1f MB algorithmic per launch = "
              "%.0f GB/s = %.1f %% of 8 TB/s" % (B, m["G"], r["Calls"], us, B * alg / 1e6, gbs, gbs / 80))
PY
done
cat $out
