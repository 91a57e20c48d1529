#!/bin/bash
# the driver's command timed, then the round's soak (shifted seeds)
cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f2_bench.json 2> gpurun_out/f2_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open('gpurun_out/f2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline'])
PY
bash tools/soak.sh 1 3 > gpurun_out/r05_soak.txt 2>&1
cat gpurun_out/r05_soak.txt
