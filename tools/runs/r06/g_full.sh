#!/bin/bash
# round 6: the whole GPU suite, then the frame pass at the three sizes
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-full}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -5 gpurun_out/r06/${T}_pytest.txt
TAG=$T bash tools/runs/r06/f.sh
