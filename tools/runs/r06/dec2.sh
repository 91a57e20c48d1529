#!/bin/bash
# round 6, run dec2: kernels of the 32 B pictures of the decoded sequence (profiling aid: a library
# variant whose intra-wave launch can be refused, so rocprofv3 does not meet a cooperative launch)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
cp xvc_amd/libxvcgpu.so /tmp/libxvcgpu_cur.so
cp variants/libxvcgpu_nocoop.so xvc_amd/libxvcgpu.so
out=/tmp/kt_dec; rm -rf $out; mkdir -p $out
cat > /tmp/dec_seq.py <<PY
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from xvc_amd import api
ctx = api.Context(0)
r = bench.stream_decode_figure(ctx, api)
print({k: v for k, v in r.items() if k in ("pictures_per_s", "ms_per_picture", "ms_by_picture_type")})
PY
( cd /tmp && XVCGPU_NO_COOPERATIVE=1 TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python /tmp/dec_seq.py > $out/log.txt 2>&1 )
tail -2 $out/log.txt | cut -c1-300
python - <<PY | tee gpurun_out/r06/dec_b_pictures_kernels.txt
import csv, glob, collections
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
print("kernels", len(rows))
# sequences: split where the host idles > 2 ms between kernels (between repetitions the pictures are checked / timed)
seqs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 1500000:
        seqs.append(cur); cur = []
    cur.append(b)
seqs.append(cur)
print("sequences", [len(s) for s in seqs])
done = 0
for s in seqs:
    ip = [i for i, r in enumerate(s) if r[2].startswith("inter_pred")]
    if len(ip) < 30: continue
    part = s[ip[0]:]
    span = (part[-1][1] - part[0][0]) / 1e3
    acc = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for st, en, n in part:
        acc[n][0] += 1; acc[n][1] += (en - st) / 1e3; busy += (en - st) / 1e3
    gaps = sum(max(0, part[i + 1][0] - part[i][1]) for i in range(len(part) - 1)) / 1e3
    print("B pictures of one sequence: span %.0f us, kernels busy %.0f us, idle between kernels %.0f us, %d launches" % (span, busy, gaps, len(part)))
    done += 1
    if done == 3:
        for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print("   %-50s calls %5d total %8.1f us avg %7.1f" % (n[:50], c, t, t / c))
        break
PY
cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so
