#!/bin/bash
# round-5 GPU check A: RDOQ parity first, then the whole GPU suite, then the quantiser's cost
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rdoq.py -x -q > gpurun_out/a_rdoq.log 2>&1
echo "rdoq tests rc=$?" | tee -a gpurun_out/a_rdoq.log
tail -25 gpurun_out/a_rdoq.log
timeout 300 python tools/rdoq_latency.py > gpurun_out/a_latency.log 2>&1
cat gpurun_out/a_latency.log
CHAIN=15 STREAMS=3 ONLY=quant_rdo timeout 600 python tools/throughput_cost.py > gpurun_out/a_tc.log 2>&1
tail -8 gpurun_out/a_tc.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/a_full.log 2>&1
echo "full gpu suite rc=$?" | tee -a gpurun_out/a_full.log
tail -15 gpurun_out/a_full.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench.log 2>&1
tail -c 3000 gpurun_out/a_bench.log
