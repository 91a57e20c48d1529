#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rdoq.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for cfg in "1920 1080 32 300 30" "3840 2160 27 300 30" "7680 4320 37 100 10"; do
  set -- $cfg
  python bench.py --width $1 --height $2 --qp $3 --steps $4 --warmup $5 --no-decode --no-cpu > gpurun_out/i_bench_$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/i_bench_$2.json") if x.startswith("{")][-1])
print($2, round(d["value"],1), d["roofline"]["all_kernels_ms"])
PY
done
