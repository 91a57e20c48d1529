#!/bin/bash
# round 6, run t: the exact-16x16 search kernel at 4 / 5 / 6 waves per SIMD (variant
# libraries under variants/, installed over the box's copy one after the other)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
cp xvc_amd/libxvcgpu.so /tmp/libxvcgpu_cur.so
for v in cur w4 w6; do
  if [ $v = cur ]; then cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so; else cp variants/libxvcgpu_$v.so xvc_amd/libxvcgpu.so || continue; fi
  echo "== $v"
  python tools/time_me.py 2>/dev/null
  out=/tmp/kt_$v; rm -rf $out; mkdir -p $out
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python $R/tools/time_me.py > $out/log.txt 2>&1 )
  python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "me_" in r["Name"]:
        print("%-70s calls %5s avg %9.1f us min %8.1f max %8.1f" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('1080p', d['value'], d['roofline']['ms_per_launch'], d['roofline']['in_flight'])"
  python bench.py --width 3840 --height 2160 --qp 27 --steps 300 --warmup 30 --no-decode --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('2160p', d['value'], d['roofline']['all_kernels_ms'])"
  python bench.py --width 7680 --height 4320 --qp 37 --steps 100 --warmup 10 --no-decode --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('4320p', d['value'], d['roofline']['all_kernels_ms'])"
done > gpurun_out/r06/t_variants.txt 2>&1
cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so
cat gpurun_out/r06/t_variants.txt
