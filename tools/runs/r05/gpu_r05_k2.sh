#!/bin/bash
# the other schedules' one-GPU points still run (sub-GOP, row shards in loop-back)
cd $GRAFT_REPO_ROOT
for args in "--schedule subgop --steps 48 --warmup 16 --no-cpu --no-decode" "--schedule rows --force-sharded --width 3840 --height 2160 --qp 32 --steps 20 --warmup 4 --no-cpu --no-decode"; do
  python bench.py --gpus 1 $args 2> gpurun_out/k2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args'.split()[1], round(d['value'], 1), d['unit'], d['config'].get('workload', '')[:80])" || tail -5 gpurun_out/k2.err
done
