#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_cu_state.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15
