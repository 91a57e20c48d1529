#!/bin/bash
# Bench lines + rocprofv3 kernel stats of the larger single-GPU BASELINE configurations
# (run on the GPU box): tools/profile_configs.sh <tag>  -> gpurun_out/profile_<tag>_configs/
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_${tag}_configs
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "2160p 3840 2160 27 300 30" "4320p 7680 4320 37 100 10"; do
  set -- $cfg
  B="python $R/bench.py --width $2 --height $3 --qp $4 --steps $5 --warmup $6 --no-decode"
  $B --cpu-frames 2 > $out/${tag}_bench_$1_qp$4.json 2> $out/bench_$1.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$1 -o st -- $B --no-cpu > $out/stats_$1.log 2>&1
  cp $(find $out/stats_$1 -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_$1_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch_$1 -o f -- $B --no-cpu --steps 30 --warmup 6 --settle 0 > $out/fetch_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write_$1 -o w -- $B --no-cpu --steps 30 --warmup 6 --settle 0 > $out/write_$1.log 2>&1
  md5=$(cd $R && python -c "import bench; print(bench.kernel_source_md5())")
  python $R/tools/pmc_traffic.py $out/fetch_$1 $out/write_$1 $out/${tag}_traffic_$1.json $md5 rdoq
  head -8 $out/${tag}_bench_$1_kernel_stats.csv | cut -c1-200
  cut -c1-700 $out/${tag}_bench_$1_qp$4.json
done
