#!/bin/bash
# round 6, run o: the search alone - time and PMC instruction counts
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-o}
python tools/time_me.py > gpurun_out/r06/${T}_time_me.txt 2>&1
cat gpurun_out/r06/${T}_time_me.txt
for f in 3; do
  bash tools/pmc_me.sh $f gpurun_out/r06/${T}_pmc_me$f > gpurun_out/r06/${T}_pmc_me$f.txt 2>&1
  grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAVE_CYCLES" gpurun_out/r06/${T}_pmc_me$f.txt
done
rm -rf gpurun_out/r06/${T}_pmc_me3 gpurun_out/r06/${T}_pmc_me*.set*.log
timeout 900 python -m pytest tests -m gpu -x -q -k "me_search or me_calls or refs_forms" > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -3 gpurun_out/r06/${T}_pytest.txt
