#!/bin/bash
# the intra picture's dependency chain with all workgroups on ONE XCD (hand-over through one L2)
cd $GRAFT_REPO_ROOT
for w in 0 16 32 48 64 96; do
  echo "== XVCGPU_INTRA_ONE_XCD=$w"
  XVCGPU_INTRA_ONE_XCD=$w python tools/time_decoder.py c1x 2>&1 | grep -v amdgpu.ids | grep " I:\|overall"
done
XVCGPU_INTRA_ONE_XCD=32 timeout 900 python -m pytest tests -m gpu -x -q -k "decod or intra or stream" 2>&1 | grep -E "passed|failed" | tail -2
