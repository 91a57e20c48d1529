#!/bin/bash
# the 16 class's search as two launches (full-pel at five waves a SIMD) against the fused kernel
cd $GRAFT_REPO_ROOT
XVCGPU_ME16_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_me_calls.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for v in 0 1; do
  for rep in 1 2; do
    XVCGPU_ME16_SPLIT=$v python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('split=$v', round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_ms']['me_search'])"
  done
  XVCGPU_ME16_SPLIT=$v python bench.py --no-cpu --no-decode --width 7680 --height 4320 --qp 37 --steps 150 --warmup 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('4320p split=$v', round(d['value']), d['ms_per_step'])"
done
