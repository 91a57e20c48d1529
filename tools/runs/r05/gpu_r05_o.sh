#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cu_state.py -x -q -s -k "engine" 2>&1 | grep -v amdgpu.ids | tail -4
for sn in 2 4; do
ENGINE_STREAMS=$sn timeout 1800 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128 --no-check 2>gpurun_out/m.err | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['chains'].items(): print('streams $sn k', k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
"
done
tail -2 gpurun_out/m.err
