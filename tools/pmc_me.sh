#!/bin/bash
# PMC passes over the ME kernel (tools/run_me_once.py <flags>); one counter set per
# rocprofv3 run (kernel-trace + pmc only).  usage: tools/pmc_me.sh <flags> <outdir> [mem]
flags=${1:-3}; out=${2:-gpurun_out/pmc_me}; which=${3:-sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$which" = "mem" ]; then
sets=("GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
      "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
      "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD"
      "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT"
      "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TD_TC_STALL_sum")
else
sets=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
      "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
      "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT")
fi
i=0
for set in "${sets[@]}"; do
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
        print(k, "n=%d" % len(v), "last=%.0f" % v[-1])
PY
