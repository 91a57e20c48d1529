#!/bin/bash
# builds and runs tools/micro/tail_time.hip (run on the GPU box): the launch time of
# deblock_tail_kernel with steps left out (TAIL_SKIP bits: 1 CU records, 2 vertical
# edges, 4 horizontal edges, 8 border, 32 own stores, 64 tile loads), then once
# with per-phase clock readings (TAIL_TRACE)
R=${GRAFT_REPO_ROOT:-.}
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I $R/xvc_amd/csrc -I $R/include"
for v in ${TAIL_VARIANTS:-0 1 2 4 8 7 15 47 111}; do
  $CC -DTAIL_SKIP=$v $R/tools/micro/tail_time.hip -o /tmp/tail_time_$v.bin 2>/dev/null && /tmp/tail_time_$v.bin
done
$CC -DTAIL_TRACE=1 $R/tools/micro/tail_time.hip -o /tmp/tail_time_tr.bin 2>/dev/null && /tmp/tail_time_tr.bin
