#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python bench.py > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline'])
e=d['config'].get('encoder_rd_serial') or d.get('encoder_rd_serial')
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
print(json.dumps(find(d,'engine'))[:1500])
PY
