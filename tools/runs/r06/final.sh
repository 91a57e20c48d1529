#!/bin/bash
# round 6, last run on the final tree: smoke, the whole GPU suite, the default bench line
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/final_pytest.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r06/final_pytest.txt | tail -2
( time python bench.py > gpurun_out/r06/final_bench.json 2> gpurun_out/r06/final_bench.err ) 2> gpurun_out/r06/final_bench_time.txt
tail -3 gpurun_out/r06/final_bench_time.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r06/final_bench.json").read().strip().split("\n")[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], {k:r[k] for k in ("kernel","frac","traffic","ms_per_launch")}, r.get("valu_issue",{}) and r["valu_issue"].get("wave_instructions"))
print(d["cpu_baseline"]["value"], d.get("stream_decode",{}).get("pictures_per_s"))
PY
