#!/bin/bash
# Produces the files that go under profiles/ for one round (run on the GPU box):
#   <tag>_bench.json               bench.py's line
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   <tag>_traffic.json             HBM bytes per launch from separate --pmc passes
# usage: tools/profile_bench.sh <tag>   (outputs in gpurun_out/profile_<tag>/)
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $out/${tag}_bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python $R/bench.py --no-cpu > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python $R/bench.py --no-cpu --steps 56 --warmup 14 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python $R/bench.py --no-cpu --steps 56 --warmup 14 > $out/write.log 2>&1
python $R/tools/pmc_traffic.py $out/fetch $out/write $out/${tag}_traffic.json
head -12 $out/${tag}_bench_kernel_stats.csv
cat $out/${tag}_bench.json
