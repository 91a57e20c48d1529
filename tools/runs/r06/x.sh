#!/bin/bash
# round 6, run x: the fused tail compiled for 3 / 4 / 5 waves per SIMD (variant libraries)
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
cp xvc_amd/libxvcgpu.so /tmp/libxvcgpu_cur.so
for v in cur t4 t5; do
  cd $R
  if [ $v = cur ]; then cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so; else cp variants/libxvcgpu_$v.so xvc_amd/libxvcgpu.so || continue; fi
  echo "== $v"
  bash tools/tail_batched.sh /tmp/tb_$v.txt | cut -c1-60,150-400
  cd $R
  python bench.py --no-cpu --no-decode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('1080p', d['value'], d['roofline']['all_kernels_ms'])"
  python bench.py --width 7680 --height 4320 --qp 37 --steps 100 --warmup 10 --no-decode --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('4320p', d['value'], d['roofline']['all_kernels_ms'])"
done > gpurun_out/r06/x_variants.txt 2>&1
cp /tmp/libxvcgpu_cur.so xvc_amd/libxvcgpu.so
cat gpurun_out/r06/x_variants.txt
