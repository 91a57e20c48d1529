#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_me_calls.py tests/test_gpu_parity.py tests/test_gpu_host_inter_search.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6
CHAIN=15 XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so python tools/trace_me.py 3 2>&1 | grep -v amdgpu.ids | head -30
CHAIN=15 python tools/throughput_cost.py 2>&1 | grep -v amdgpu.ids | tail -25
