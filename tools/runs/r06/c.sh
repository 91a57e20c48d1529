#!/bin/bash
# round 6, run c: parity of the searches after the full-pel prologue rework, instruction counts
# (PMC) of the search by phase, phase clocks
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "me_search or me_calls or refs_forms or host_inter_search or frame_pass or smoke" > gpurun_out/r06/c_pytest.txt 2>&1
tail -5 gpurun_out/r06/c_pytest.txt
python tools/time_me.py > gpurun_out/r06/c_time_me.txt 2>&1
cat gpurun_out/r06/c_time_me.txt
for f in 1 2 3; do
  bash tools/pmc_me.sh $f gpurun_out/r06/c_pmc_me$f > gpurun_out/r06/c_pmc_me$f.txt 2>&1
  grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_THREAD_CYCLES_VALU|SQ_WAIT_INST_LDS|SQ_LDS_BANK" gpurun_out/r06/c_pmc_me$f.txt
done
rm -rf gpurun_out/r06/c_pmc_me1 gpurun_out/r06/c_pmc_me2 gpurun_out/r06/c_pmc_me3
bash tools/trace_me.sh 3 > gpurun_out/r06/c_trace_me.txt 2>&1
cat gpurun_out/r06/c_trace_me.txt | cut -c1-200
