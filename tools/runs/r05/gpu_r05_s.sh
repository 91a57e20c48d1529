#!/bin/bash
cd $GRAFT_REPO_ROOT
CHAIN=15 XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so python tools/trace_me.py 3 2>&1 | grep -v amdgpu.ids | head -30
python tools/me_phase_counts.py 2>&1 | grep -v amdgpu.ids | tail -30
