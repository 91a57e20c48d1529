#!/bin/bash
# round 6, run q: the quantiser's parity tests, then the frame pass at the three sizes
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-q}
timeout 1500 python -m pytest tests -m gpu -x -q -k "rdoq or frame_pass or residual or c1_decision or rd_calls" > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -3 gpurun_out/r06/${T}_pytest.txt
TAG=$T bash tools/runs/r06/f.sh
