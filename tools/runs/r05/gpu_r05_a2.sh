#!/bin/bash
# the driver's short run (--steps 20 --warmup 5) against the chain count and the step count
cd $GRAFT_REPO_ROOT
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --chains 2" "--steps 20 --warmup 5 --chains 4" "--steps 21 --warmup 6" "--steps 20 --warmup 5 --graph" "--steps 200 --warmup 20"; do
  for rep in 1 2; do
    python bench.py --no-cpu --no-decode $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', round(d['value']), d['ms_per_step'], d['config'].get('chains'))"
  done
done
