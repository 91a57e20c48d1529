#!/bin/bash
# PMC counters of one frame-pass kernel: tools/pmc_kernel.sh <me|recon|deblock|pad|ssd> <kernel-name-substring>
which=${1:-recon}; pat=${2:-recon_from_me}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_$which
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/set$i -o pmc -- python $R/tools/time_kernels.py $which > $out/set$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/set*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "$pat" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k, "n=%d" % len(v), "last=%.0f" % v[-1])
PY
