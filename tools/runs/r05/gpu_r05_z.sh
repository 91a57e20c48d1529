#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python bench.py > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err; echo bench rc=$?
tail -3 gpurun_out/z_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/z_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline'])
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
print(json.dumps(find(d,'lic_picture'))[:1500])
print(json.dumps(find(d,'engine'))[:600])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
