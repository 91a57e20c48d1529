#!/bin/bash
# the round's last run on the final sources: the GPU suite, the bench profile (traffic / issue
# counters tied to the kernel sources' MD5), the CU-state walks on the representative stretch
R=$GRAFT_REPO_ROOT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/final_pytest.txt
cat gpurun_out/final_pytest.txt
bash tools/profile_bench.sh r05 rdoq > gpurun_out/profile_r05.log 2>&1
cd /tmp && export TMPDIR=/tmp
for m in chained live serial; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_walk_$m -o walk -- python $R/tools/cu_state_walk.py --mode $m --states 3200 --k 1 --no-check > /dev/null 2>&1
  cp $(find $R/gpurun_out/r05_walk_$m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_cu_state_${m}_kernel_stats.csv
done
ENGINE_THREADS=4 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_walk_engine -o walk -- python $R/tools/cu_state_walk.py --mode engine --states 1500 --k 128 --no-check > $R/gpurun_out/r05_walk_engine_prof.json 2>/dev/null
cp $(find $R/gpurun_out/r05_walk_engine -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_cu_state_engine_kernel_stats.csv
cd $R
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128,256 > gpurun_out/r05_walk_engine.json 2> gpurun_out/r05_walk_engine.err
ENGINE_THREADS=1 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,128 --no-check > gpurun_out/r05_walk_engine_one_thread.json 2>> gpurun_out/r05_walk_engine.err
for m in serial chained live; do
  python tools/cu_state_walk.py --mode $m --states 4000 --k 1,4,8 > gpurun_out/r05_walk_$m.json 2> gpurun_out/r05_walk_$m.err
done
python tools/cu_state_walk.py --clip tiny --mode chained --states 100000 --k 1 > gpurun_out/r05_walk_tiny_chained.json 2> gpurun_out/r05_walk_tiny.err
python tools/cu_state_walk.py --clip tiny --mode live --states 100000 --k 1 > gpurun_out/r05_walk_tiny_live.json 2>> gpurun_out/r05_walk_tiny.err
python tools/cu_state_walk.py --clip tiny --mode serial --states 100000 --k 1 > gpurun_out/r05_walk_tiny_serial.json 2>> gpurun_out/r05_walk_tiny.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -rf gpurun_out/profile_r05/stats gpurun_out/profile_r05/fetch gpurun_out/profile_r05/write gpurun_out/profile_r05/sq[0-9]*
rm -rf gpurun_out/r05_walk_chained gpurun_out/r05_walk_live gpurun_out/r05_walk_serial gpurun_out/r05_walk_engine
tail -3 gpurun_out/profile_r05.log | cut -c1-400
