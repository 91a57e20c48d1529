#!/bin/bash
# round-5 profiles (run on the GPU box): everything that goes under profiles/r05_*
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_bench.sh r05 rdoq > gpurun_out/profile_r05.log 2>&1
bash tools/profile_configs.sh r05 > gpurun_out/profile_r05_configs.log 2>&1
CHAIN=15 python tools/throughput_cost.py > gpurun_out/r05_throughput_cost.txt 2>&1
STATE=steady CHAIN=15 XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so python tools/trace_rdoq.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_rdoq_sections_steady.txt
python tools/rdoq_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_rdoq_latency.txt
cd /tmp && export TMPDIR=/tmp
for m in chained live serial; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_walk_$m -o walk -- python $R/tools/cu_state_walk.py --mode $m --states 3200 --k 1 --no-check > /dev/null 2>&1
  cp $(find $R/gpurun_out/r05_walk_$m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_cu_state_${m}_kernel_stats.csv
done
ENGINE_THREADS=4 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_walk_engine -o walk -- python $R/tools/cu_state_walk.py --mode engine --states 1500 --k 128 --no-check > /dev/null 2>&1
cp $(find $R/gpurun_out/r05_walk_engine -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05_cu_state_engine_kernel_stats.csv
cd $R
ENGINE_THREADS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,48,128,256 > gpurun_out/r05_walk_engine.json 2> gpurun_out/r05_walk_engine.err
ENGINE_THREADS=1 python tools/cu_state_walk.py --mode engine --states 2500 --k 16,128 --no-check > gpurun_out/r05_walk_engine_one_thread.json 2>> gpurun_out/r05_walk_engine.err
for m in chained live; do
  python tools/cu_state_walk.py --mode $m --states 4000 --k 1,4,8 > gpurun_out/r05_walk_$m.json 2> gpurun_out/r05_walk_$m.err
done
python tools/tail_hbm.py > gpurun_out/r05_tail_hbm.txt 2>&1
# the raw traces stay on the box (gpurun merges at most 64 MiB back): summaries only
rm -rf gpurun_out/profile_r05/stats gpurun_out/profile_r05/fetch gpurun_out/profile_r05/write gpurun_out/profile_r05/sq[0-9]*
rm -rf gpurun_out/profile_r05_configs/stats_* gpurun_out/profile_r05_configs/fetch_* gpurun_out/profile_r05_configs/write_*
find gpurun_out/profile_r05_configs gpurun_out/profile_r05 -type d -mindepth 1 -exec rm -rf {} + 2>/dev/null
rm -rf gpurun_out/r05_walk_chained gpurun_out/r05_walk_live gpurun_out/r05_walk_serial gpurun_out/r05_walk_engine
du -sh gpurun_out; ls gpurun_out/profile_r05 gpurun_out/profile_r05_configs
