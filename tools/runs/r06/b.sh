#!/bin/bash
# round 6, run b: the reworked sub-pel sweep + full-pel addressing: parity of every test that
# reaches the motion search, then the search alone
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "me_search or me_calls or refs_forms or host_inter_search or frame_pass or smoke or cu_state" > gpurun_out/r06/b_pytest.txt 2>&1
tail -15 gpurun_out/r06/b_pytest.txt
python tools/time_me.py > gpurun_out/r06/b_time_me.txt 2>&1
cat gpurun_out/r06/b_time_me.txt
