#!/bin/bash
# the class lists as one launch (rdoq_lists_kernel) against the two launches before
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_rdoq.py tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for v in 0 1; do
  for rep in 1 2; do
    XVCGPU_RDOQ_TWO_LAUNCH_LISTS=$v python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two_launch=$v', round(d['value']), d['ms_per_step'])"
  done
done
for v in 0 1; do
  XVCGPU_RDOQ_TWO_LAUNCH_LISTS=$v python bench.py --no-cpu --no-decode --width 3840 --height 2160 --qp 27 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('2160p two_launch=$v', round(d['value']), d['ms_per_step'])"
  XVCGPU_RDOQ_TWO_LAUNCH_LISTS=$v python bench.py --no-cpu --no-decode --width 7680 --height 4320 --qp 37 --steps 150 --warmup 15 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('4320p two_launch=$v', round(d['value']), d['ms_per_step'])"
done
