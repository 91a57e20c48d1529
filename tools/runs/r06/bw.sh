#!/bin/bash
# round 6, run bw: the refinement search's workgroup at 8 waves (32 / 64 class), variant library
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
cp xvc_amd/libxvcgpu.so /tmp/libxvcgpu_cur.so
for v in cur b8; do
  if [ $v = cur ]; then cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so; else cp variants/libxvcgpu_$v.so xvc_amd/libxvcgpu.so || continue; fi
  echo "== $v"
  if [ $v != cur ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "bipred or rd_calls or refs_forms or engine" 2>&1 | tail -2; fi
  ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['chains'].items(): print('engine k', k, round(v['pictures_per_s'],3), 'pictures/s', round(v['us_per_cu_state_aggregate'],2), 'us/state')"
  python tools/cu_state_walk.py --mode chained --states 4000 --k 1 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['chains'].items(): print('chained k', k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,list))})" | cut -c1-400
done > gpurun_out/r06/bw_variants.txt 2>&1
cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so
cat gpurun_out/r06/bw_variants.txt
