#!/bin/bash
# the engine's figures and kernel stats (profiles/r05_cu_state_walk_engine.json, *_engine_kernel_stats.csv)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ENGINE_THREADS=4 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_walk_engine -o walk -- python $R/tools/cu_state_walk.py --mode engine --states 1500 --k 128 --no-check > $R/gpurun_out/r05_walk_engine_prof.json 2>/dev/null
cp $(find $R/gpurun_out/r05_walk_engine -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_cu_state_engine_kernel_stats.csv
rm -rf $R/gpurun_out/r05_walk_engine
cd $R
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128,256 > gpurun_out/r05_walk_engine.json 2> gpurun_out/r05_walk_engine.err
ENGINE_THREADS=1 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,128 --no-check > gpurun_out/r05_walk_engine_one_thread.json 2>> gpurun_out/r05_walk_engine.err
tail -c 1500 gpurun_out/r05_walk_engine.json
python tools/me_phase_counts.py 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/u_me_phase_counts.txt
cat gpurun_out/u_me_phase_counts.txt
