#!/bin/bash
# round 6, run z2: the decoder's kernels in the pipelined sequence mode (bench's stream_decode leg)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
out=/tmp/kt_dec; rm -rf $out; mkdir -p $out
cat > /tmp/dec_seq.py <<PY
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np
import bench
from xvc_amd import api
ctx = api.Context(0)
t0 = time.perf_counter()
r = bench.stream_decode_figure(ctx, api)
print({k: v for k, v in r.items() if k in ("pictures_per_s", "ms_per_picture", "ms_by_picture_type")})
PY
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python /tmp/dec_seq.py > $out/log.txt 2>&1 )
tail -3 $out/log.txt
python - <<PY | tee gpurun_out/r06/z_decoder_kernels.txt
import csv, glob, collections
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
# sequence runs: find I-picture kernels (intra_waves) as markers; take the sequence runs 2..6 (pipelined reps)
marks = [i for i, r in enumerate(rows) if r[2].startswith("intra_waves")]
print("kernels", len(rows), "intra_waves launches", len(marks))
# group consecutive marks
starts = [marks[0]] + [marks[i] for i in range(1, len(marks)) if rows[marks[i]][0] - rows[marks[i-1]][1] > 2000000]
print("sequence starts", len(starts))
for s in range(1, min(6, len(starts) - 1)):
    a, b = starts[s], starts[s + 1]
    seg = rows[a:b]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    acc = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for st, en, n in seg:
        acc[n][0] += 1; acc[n][1] += (en - st) / 1e3; busy += (en - st) / 1e3
    print("sequence %d: span %.2f ms, kernels busy %.2f ms, %d launches" % (s, span, busy / 1e3, len(seg)))
    if s == 2:
        for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print("   %-56s calls %5d total %8.1f us avg %7.1f" % (n[:56], c, t, t / c))
        # the I picture's end = last intra kernel before first inter_pred
        first_inter = next(i for i, r in enumerate(seg) if r[2].startswith("inter_pred"))
        print("   I picture: %.2f ms until the first inter_pred" % ((seg[first_inter][0] - seg[0][0]) / 1e6))
        gaps = [seg[i + 1][0] - seg[i][1] for i in range(first_inter, len(seg) - 1)]
        print("   after it: %d launches, idle between kernels total %.2f ms" % (len(gaps), sum(g for g in gaps if g > 0) / 1e6))
PY
