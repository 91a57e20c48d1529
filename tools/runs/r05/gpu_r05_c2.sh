#!/bin/bash
# me_search bounded to N workgroups (three waves a SIMD = 3072) against a workgroup per job
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_me_calls.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for cap in 3072 0 2048 4096 3584; do
  for rep in 1 2; do
    XVCGPU_ME_WAVES_CAP=$cap python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap=$cap', round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_ms']['me_search'])"
  done
done
for cap in 3072 0; do
  XVCGPU_ME_WAVES_CAP=$cap python bench.py --no-cpu --no-decode --width 3840 --height 2160 --qp 27 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('2160p cap=$cap', round(d['value']), d['ms_per_step'])"
  XVCGPU_ME_WAVES_CAP=$cap python bench.py --no-cpu --no-decode --width 7680 --height 4320 --qp 37 --steps 150 --warmup 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('4320p cap=$cap', round(d['value']), d['ms_per_step'])"
done
