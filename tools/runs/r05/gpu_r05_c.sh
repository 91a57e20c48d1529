#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CHAIN=${CHAIN:-15} XVCGPU_LIB=$PWD/xvc_amd/libxvcgpu_trace.so timeout 600 python tools/trace_rdoq.py 2>&1 | grep -v amdgpu.ids | head -8
exit 0
CHAIN=15 REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c_prof -o rq -- python $GRAFT_REPO_ROOT/tools/run_rdoq_steady.py > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/c_prof/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "quant_rdo_packed" in r["Kernel_Name"]]
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
    print("quant_rdo_packed_kernel: %d launches, last 20: mean %.1f us min %.1f max %.1f" % (len(d), sum(d[-20:])/20, min(d[-20:]), max(d[-20:])))
    print("grid", rows[-1].get("Grid_Size_X"), "wg", rows[-1].get("Workgroup_Size_X"), "lds", rows[-1].get("LDS_Block_Size"), "vgpr", rows[-1].get("VGPR_Count"))
PY
