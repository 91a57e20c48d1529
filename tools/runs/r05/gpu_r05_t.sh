#!/bin/bash
cd $GRAFT_REPO_ROOT
for hnd in 0 1 2; do
echo "== XVCGPU_INTRA_HAND=$hnd"
XVCGPU_INTRA_HAND=$hnd python tools/time_decoder.py c1 2>&1 | grep -v amdgpu.ids | grep " I:\|overall"
XVCGPU_INTRA_HAND=$hnd python tools/time_decoder.py c1x 2>&1 | grep -v amdgpu.ids | grep " I:\|overall"
done
for i in 1 2 3; do
XVCGPU_INTRA_HAND=2 timeout 900 python -m pytest tests -m gpu -x -q -k "decod or intra" 2>&1 | grep -v amdgpu.ids | tail -2
done
