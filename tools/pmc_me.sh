#!/bin/bash
# PMC passes over the ME kernel (tools/run_me_once.py <flags>); one counter set per run.
# usage: tools/pmc_me.sh <flags> <outdir>
flags=${1:-3}; out=${2:-gpurun_out/pmc_me}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/set$i -o pmc -- python $R/tools/run_me_once.py $flags > $R/$out.set$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/$out/set*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "me_search" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k, "n=%d" % len(v), "last=%.0f" % v[-1], "mean=%.0f" % (sum(v) / len(v)))
PY
