R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/tl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python $R/bench.py --no-cpu --no-decode --steps 120 --warmup 30 > $out/log.txt 2>&1
python $R/tools/timeline.py $(find $out -name "*kernel_trace.csv" | head -1)
