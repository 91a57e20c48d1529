#!/bin/bash
# round 6, run e: phase and sub-pel section clocks of the search
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
bash tools/trace_me.sh 3 > gpurun_out/r06/${TAG:-e}_trace_me.txt 2>&1
cat gpurun_out/r06/${TAG:-e}_trace_me.txt | cut -c1-200
