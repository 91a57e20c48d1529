#!/bin/bash
# SQ / LDS counters of quant_rdo_packed_kernel on the chain's steady state
# (run on the GPU box): tools/pmc_rdoq.sh [tag]  -> gpurun_out/pmc_rdoq_<tag>.txt
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_rdoq_$tag
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -o pmc -- python $R/tools/run_rdoq_steady.py > $out/set$i.log 2>&1
done
python - > $R/gpurun_out/pmc_rdoq_$tag.txt <<PY
import csv, glob, collections
print("# quant_rdo_packed_kernel, 1080p QP ${QP:-32}, chain steady state (CHAIN=${CHAIN:-120}); last launch of each pass")
for f in sorted(glob.glob("$out/set*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "quant_rdo_packed" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s n=%d last=%.0f" % (k, len(v), v[-1]))
PY
cat $R/gpurun_out/pmc_rdoq_$tag.txt
