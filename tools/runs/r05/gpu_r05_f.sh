#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rdoq.py -x -q 2>&1 | tail -12
timeout 300 python tools/rdoq_latency.py 2>&1 | grep -v amdgpu.ids
CHAIN=15 STREAMS=3 ONLY=quant_rdo timeout 600 python tools/throughput_cost.py 2>&1 | tail -3
STATE=steady CHAIN=15 XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so timeout 600 python tools/trace_rdoq.py 2>&1 | grep -v amdgpu.ids | tail -16
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/f_full.log 2>&1
echo "full gpu suite rc=$?"; tail -12 gpurun_out/f_full.log
timeout 900 python bench.py --steps 300 --warmup 30 --no-cpu > gpurun_out/f_bench.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/f_bench.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms"])
e=d["stream_decode"]["encoder_rd_serial"]; print("serial us/state", e["us_per_cu_state"], "chained", e["chained"]["us_per_cu_state"], e["chained"]["pictures_per_s"], e["pictures_per_s"])
print("decode", d["stream_decode"].get("pictures_per_s"))
PY
