#!/bin/bash
# round 6, run d: search alone (time, PMC instruction counts by phase, phase clocks); parity of the
# searches last
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-d}
python tools/time_me.py > gpurun_out/r06/${T}_time_me.txt 2>&1
cat gpurun_out/r06/${T}_time_me.txt
for f in 1 2 3; do
  bash tools/pmc_me.sh $f gpurun_out/r06/${T}_pmc_me$f > gpurun_out/r06/${T}_pmc_me$f.txt 2>&1
  grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_THREAD_CYCLES_VALU|SQ_WAIT_INST_LDS|SQ_LDS_BANK" gpurun_out/r06/${T}_pmc_me$f.txt
done
rm -rf gpurun_out/r06/${T}_pmc_me1 gpurun_out/r06/${T}_pmc_me2 gpurun_out/r06/${T}_pmc_me3 gpurun_out/r06/${T}_pmc_me*.set*.log
bash tools/trace_me.sh 3 > gpurun_out/r06/${T}_trace_me.txt 2>&1
cat gpurun_out/r06/${T}_trace_me.txt | cut -c1-200 | head -14
timeout 1500 python -m pytest tests -m gpu -x -q -k "me_search or me_calls or refs_forms or host_inter_search or frame_pass or smoke" > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -5 gpurun_out/r06/${T}_pytest.txt
