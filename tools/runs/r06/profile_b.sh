#!/bin/bash
# round 6 profiles, part b (run on the GPU box): the CU-state walks - kernel stats per form,
# the figures per k
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for m in chained live serial; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_walk_$m -o walk -- python $R/tools/cu_state_walk.py --mode $m --states 3200 --k 1 --no-check > /dev/null 2>&1
  cp $(find /tmp/r06_walk_$m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/r06_cu_state_${m}_kernel_stats.csv
done
ENGINE_THREADS=4 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_walk_engine -o walk -- python $R/tools/cu_state_walk.py --mode engine --states 1500 --k 128 --no-check > /dev/null 2>&1
cp $(find /tmp/r06_walk_engine -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/r06_cu_state_engine_kernel_stats.csv
cd $R
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128,256 > gpurun_out/r06/r06_cu_state_walk_engine.json 2> gpurun_out/r06/walk_engine.err
ENGINE_THREADS=1 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,128 --no-check > gpurun_out/r06/r06_cu_state_walk_engine_one_thread.json 2>> gpurun_out/r06/walk_engine.err
for m in serial chained live; do
  python tools/cu_state_walk.py --mode $m --states 4000 --k 1,4,8 > gpurun_out/r06/r06_cu_state_walk_$m.json 2> gpurun_out/r06/walk_$m.err
done
for m in chained live serial; do
  python tools/cu_state_walk.py --clip tiny --mode $m --states 100000 --k 1 > gpurun_out/r06/r06_cu_state_walk_tiny_$m.json 2>> gpurun_out/r06/walk_tiny.err
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ls -la gpurun_out/r06 | tail -30
