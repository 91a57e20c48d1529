#!/bin/bash
# round 6, run l: the reference encoder's CPU time on the walk's clip, on the GPU box's host
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
nproc > gpurun_out/r06/l_nproc.txt; lscpu | head -20 >> gpurun_out/r06/l_nproc.txt
timeout 1500 python tools/ref_encoder_time.py > gpurun_out/r06/r06_ref_encoder_cpu.json 2> gpurun_out/r06/l_err.txt
cat gpurun_out/r06/r06_ref_encoder_cpu.json; tail -3 gpurun_out/r06/l_err.txt
