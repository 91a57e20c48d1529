#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/h2_bench.json 2> gpurun_out/h2_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open('gpurun_out/h2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic'))
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
print(find(d,'seconds_spent_measuring'))
print(json.dumps(find(d,'engine'))[:400])
PY
