#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/dbg/lic_walk.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 1500 python -m pytest tests/test_gpu_cu_state.py tests/test_gpu_refs_forms.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12
