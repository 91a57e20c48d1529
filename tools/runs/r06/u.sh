#!/bin/bash
# round 6, run u: the sq16-hint kernel (16x16 + 16x8 exact instances, any-size under the cap)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
T=${TAG:-u}
python tools/time_me.py 2>/dev/null | tee gpurun_out/r06/${T}_time_me.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "me_search or me_calls or refs_forms or frame_pass" > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -3 gpurun_out/r06/${T}_pytest.txt
TAG=$T bash tools/runs/r06/f.sh
