#!/bin/bash
# Produces the files that go under profiles/ for one round (run on the GPU box):
#   <tag>_bench.json               bench.py's line
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   <tag>_traffic.json             HBM bytes per launch from separate --pmc passes, stamped with
#                                  the MD5 of the kernel sources (copy to profiles/traffic_current.json, the QuantFast one to traffic_current_fast.json)
#   <tag>_sq_counters.txt          SQ / TCP / LDS counters of the pass's kernels (separate passes)
#   <tag>_issue.json               wave instructions per launch and kernel (copy to profiles/issue_current.json:
#                                  bench.py's roofline.valu_issue)
# usage: tools/profile_bench.sh <tag> [rdoq|fast]  (outputs in gpurun_out/profile_<tag>/)
tag=${1:-r02}
quant=${2:-rdoq}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --quant $quant"
$B > $out/${tag}_bench.json 2> $out/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- $B --no-cpu --no-decode > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
# (a short settle phase: under the counters every dispatch is serialised)
S="--no-cpu --no-decode --steps 56 --warmup 14 --settle 240"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- $B $S > $out/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- $B $S > $out/write.log 2>&1
md5=$(cd $R && python -c "import bench; print(bench.kernel_source_md5())")
python $R/tools/pmc_traffic.py $out/fetch $out/write $out/${tag}_traffic.json $md5 $quant
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/sq$i -o pmc -- $B $S > $out/sq$i.log 2>&1
done
python $R/tools/pmc_issue.py $out $out/${tag}_issue.json $md5 $quant
python - > $out/${tag}_sq_counters.txt <<PY
import csv, glob, collections
print("# per-launch averages over python bench.py --quant $quant $S (one rocprofv3 --pmc pass per counter set)")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$out/sq*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-36s n=%-5d avg=%.1f" % (c, len(v), sum(v) / len(v)))
PY
head -14 $out/${tag}_bench_kernel_stats.csv
cat $out/${tag}_bench.json
