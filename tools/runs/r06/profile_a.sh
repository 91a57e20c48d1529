#!/bin/bash
# round 6 profiles, part a (run on the GPU box): the reference encoder's CPU time, the bench
# profile (kernel stats, HBM traffic and instruction counters tied to the kernel sources'
# MD5), the larger configurations, the tail's HBM measurements.  Summaries only come back.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06
python tools/ref_encoder_time.py > gpurun_out/r06/r06_ref_encoder_cpu.json 2> gpurun_out/r06/ref_encoder.err
cp gpurun_out/r06/r06_ref_encoder_cpu.json profiles/r06_ref_encoder_cpu.json
bash tools/profile_bench.sh r06 rdoq > gpurun_out/profile_r06.log 2>&1
cp gpurun_out/profile_r06/r06_traffic.json profiles/traffic_current.json
cp gpurun_out/profile_r06/r06_issue.json profiles/issue_current.json
# the bench line again, now that the counter files of these kernel sources exist
python bench.py > gpurun_out/profile_r06/r06_bench.json 2> gpurun_out/profile_r06/bench.err
bash tools/profile_configs.sh r06 > gpurun_out/profile_r06_configs.log 2>&1
cd $R
python tools/tail_hbm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/r06_tail_hbm.txt
bash tools/tail_batched.sh $R/gpurun_out/r06/r06_tail_batched.txt > /dev/null 2>&1
cd $R
CHAIN=15 python tools/throughput_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/r06_throughput_cost.txt
rm -rf gpurun_out/profile_r06/stats gpurun_out/profile_r06/fetch gpurun_out/profile_r06/write gpurun_out/profile_r06/sq[0-9]*
rm -rf gpurun_out/profile_r06_configs/stats_* gpurun_out/profile_r06_configs/fetch_* gpurun_out/profile_r06_configs/write_*
find gpurun_out/profile_r06_configs gpurun_out/profile_r06 -type d -mindepth 1 -exec rm -rf {} + 2>/dev/null
du -sh gpurun_out; ls gpurun_out/profile_r06 gpurun_out/profile_r06_configs gpurun_out/r06
tail -2 gpurun_out/profile_r06.log | cut -c1-600
