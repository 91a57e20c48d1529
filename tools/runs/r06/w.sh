#!/bin/bash
# round 6, run w: batched tail measurement + the whole GPU suite on the sources of d436690
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
bash tools/tail_batched.sh $R/gpurun_out/r06/w_tail_batched.txt
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/w_pytest.txt 2>&1
tail -5 gpurun_out/r06/w_pytest.txt
