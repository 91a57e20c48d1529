#!/bin/bash
# round 6, run dec: the decoder's sequence mode with its own trace (where the issuing thread waits)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
cat > /tmp/dec_seq.py <<PY
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from xvc_amd import api
ctx = api.Context(0)
r = bench.stream_decode_figure(ctx, api)
print({k: v for k, v in r.items() if k in ("pictures_per_s", "ms_per_picture", "ms_by_picture_type")})
PY
XVC_DEC_TRACE=1 python /tmp/dec_seq.py > gpurun_out/r06/dec_trace.txt 2>&1
grep -E "DecodeSequence|pictures_per_s" gpurun_out/r06/dec_trace.txt | tail -8
