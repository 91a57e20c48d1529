#!/bin/bash
# round 6, run a: baseline of the motion search alone + its phase clocks / event counts
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
python $R/tools/time_me.py > $R/gpurun_out/r06/a_time_me.txt 2>&1
bash $R/tools/trace_me.sh 3 > $R/gpurun_out/r06/a_trace_me.txt 2>&1
CHAIN=12 bash $R/tools/trace_me.sh 3 > $R/gpurun_out/r06/a_trace_me_chain12.txt 2>&1
tail -30 $R/gpurun_out/r06/a_time_me.txt $R/gpurun_out/r06/a_trace_me.txt $R/gpurun_out/r06/a_trace_me_chain12.txt
