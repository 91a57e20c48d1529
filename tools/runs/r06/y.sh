#!/bin/bash
# round 6, run y: tail at 4 waves (main library) parity subset; RDOQ lane-per-sub-block kernel at 3 waves per SIMD (variant)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -x -q -k "deblock or tail or frame_pass or stream or picture_parallel" > gpurun_out/r06/y_pytest.txt 2>&1
tail -3 gpurun_out/r06/y_pytest.txt
cp xvc_amd/libxvcgpu.so /tmp/libxvcgpu_cur.so
for v in cur q3; do
  if [ $v = cur ]; then cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so; else cp variants/libxvcgpu_$v.so xvc_amd/libxvcgpu.so || continue; fi
  echo "== $v"
  python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('1080p', d['value'], d['roofline']['all_kernels_ms'])"
  python bench.py --width 3840 --height 2160 --qp 27 --steps 300 --warmup 30 --no-decode --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('2160p', d['value'], d['roofline']['all_kernels_ms'])"
done > gpurun_out/r06/y_variants.txt 2>&1
cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so
cat gpurun_out/r06/y_variants.txt
