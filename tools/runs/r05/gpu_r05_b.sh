#!/bin/bash
# round-5 GPU check B: section clocks of the RDOQ walk on the chain's settled state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CHAIN=${CHAIN:-15} XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so timeout 600 python tools/trace_rdoq.py > gpurun_out/b_trace.log 2>&1
cat gpurun_out/b_trace.log | tail -30
CHAIN=15 timeout 600 python tools/rdoq_sparsity.py > gpurun_out/b_sparsity.log 2>&1
tail -12 gpurun_out/b_sparsity.log
timeout 300 python -m pytest tests/test_gpu_rdoq.py -x -q 2>&1 | tail -3
CHAIN=15 STREAMS=3 ONLY=quant_rdo timeout 600 python tools/throughput_cost.py 2>&1 | tail -4
