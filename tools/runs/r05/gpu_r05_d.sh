#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rdoq.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
CHAIN=15 STREAMS=3 ONLY=quant_rdo timeout 600 python tools/throughput_cost.py 2>&1 | tail -4
CHAIN=${CHAIN:-15} XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so timeout 600 python tools/trace_rdoq.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/d_trace.log
timeout 300 python tools/rdoq_latency.py 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu --no-decode > gpurun_out/d_bench.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/d_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms"])
PY
