#!/bin/bash
# round 6, run m: the CU-state walk through the C++ composer and the event-ordered engine:
# parity of every form, then the engine's rates
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r06
cd $R
T=${TAG:-m}
timeout 2000 python -m pytest tests/test_gpu_cu_state.py tests/test_gpu_intra_calls.py -x -q -m gpu > gpurun_out/r06/${T}_pytest.txt 2>&1
tail -5 gpurun_out/r06/${T}_pytest.txt
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128 > gpurun_out/r06/${T}_walk_engine.json 2> gpurun_out/r06/${T}_walk_engine.err
tail -c 1200 gpurun_out/r06/${T}_walk_engine.json; tail -3 gpurun_out/r06/${T}_walk_engine.err
ENGINE_THREADS=1 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48 --no-check > gpurun_out/r06/${T}_walk_engine_1t.json 2>> gpurun_out/r06/${T}_walk_engine.err
tail -c 600 gpurun_out/r06/${T}_walk_engine_1t.json
