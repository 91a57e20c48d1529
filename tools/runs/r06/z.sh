#!/bin/bash
# round 6, run z: the decoder per picture (c1x) and its kernels
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
python tools/time_decoder.py c1x > gpurun_out/r06/z_time_decoder.txt 2>&1
tail -40 gpurun_out/r06/z_time_decoder.txt
out=/tmp/kt_dec; rm -rf $out; mkdir -p $out
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python $R/tools/time_decoder.py c1x > $out/log.txt 2>&1 )
python - <<PY | tee gpurun_out/r06/z_decoder_kernels.txt
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    print("%-64s calls %6s avg %9.1f us total %9.2f ms" % (r["Name"].split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("all kernels %.2f ms over 3 repetitions of 33 pictures" % (tot / 1e6))
PY
