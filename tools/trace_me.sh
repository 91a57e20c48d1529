#!/bin/bash
# developer build with the ME phase timestamps + tools/trace_me.py (run on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DXVCGPU_TRACE -I $R/include \
  $R/xvc_amd/csrc/xvcgpu.hip $R/xvc_amd/csrc/xvcgpu_comm.hip $R/xvc_amd/csrc/xvcgpu_tables.cpp \
  -o /tmp/libxvcgpu_trace.so 2>/dev/null && XVCGPU_LIB=/tmp/libxvcgpu_trace.so python $R/tools/trace_me.py "$@"
