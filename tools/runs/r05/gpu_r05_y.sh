#!/bin/bash
# the engine at k = 16 / 32: engines (host threads, a stream each) against chains per engine
cd $GRAFT_REPO_ROOT
for T in 2 4 8 16; do
  echo "== ENGINE_THREADS=$T"
  ENGINE_THREADS=$T python tools/cu_state_walk.py --mode engine --states 2500 --k 16,32 --no-check 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d['chains'].items(): print(k, round(v['pictures_per_s'], 3), round(v['launches_per_state'], 2), v['host_threads'], v['streams'])"
done
echo "== one engine, four streams"
ENGINE_THREADS=1 ENGINE_STREAMS=4 python tools/cu_state_walk.py --mode engine --states 2500 --k 16 --no-check 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d['chains'].items(): print(k, round(v['pictures_per_s'], 3), round(v['launches_per_state'], 2), v['host_threads'], v['streams'])"
