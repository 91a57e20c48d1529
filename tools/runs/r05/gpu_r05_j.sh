#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rdoq.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for lib in libxvcgpu.so libxvcgpu_w4.so; do
for cfg in "1920 1080 32 300 30" "3840 2160 27 300 30" "7680 4320 37 100 10"; do
  set -- $cfg
  XVCGPU_LIB=$PWD/xvc_amd/$lib python bench.py --width $1 --height $2 --qp $3 --steps $4 --warmup $5 --no-decode --no-cpu > gpurun_out/j_bench.json 2>/dev/null
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/j_bench.json") if x.startswith("{")][-1])
print("$lib", $2, round(d["value"],1), d["roofline"]["all_kernels_ms"]["quant_rdo"])
PY
done
done
