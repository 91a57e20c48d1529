#!/bin/bash
# round 6, run f: where the frame pass stands (1080p / 2160p / 4320p), quick lines (no CPU leg,
# no decode leg)
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-f}
python bench.py --no-cpu --no-decode > gpurun_out/r06/${T}_bench_1080p.json 2> gpurun_out/r06/${T}_bench_1080p.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06/${T}_bench_1080p.json").read().strip().split("\n")[-1])
print("1080p", d["value"], d["ms_per_step"], d.get("roofline"), {k: v for k, v in d.items() if "kernel" in k})
PY
python bench.py --width 3840 --height 2160 --qp 27 --steps 300 --warmup 30 --no-decode --no-cpu > gpurun_out/r06/${T}_bench_2160p.json 2>/dev/null
python bench.py --width 7680 --height 4320 --qp 37 --steps 100 --warmup 10 --no-decode --no-cpu > gpurun_out/r06/${T}_bench_4320p.json 2>/dev/null
python - <<PY
import json
for n in ("2160p", "4320p"):
    d=json.loads(open("gpurun_out/r06/${T}_bench_%s.json" % n).read().strip().split("\n")[-1])
    print(n, d["value"], d["ms_per_step"], d.get("roofline"))
PY
