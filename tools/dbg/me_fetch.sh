cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 0 1; do for f in 1 2 3; do
  XCD_TILES=$t rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/mef/t${t}f$f -o p -- python $R/tools/run_me_once.py $f > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/mef/t${t}f$f/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "me_search" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("tiles=$t flags=$f FETCH_SIZE KB per launch (x2 = bytes on gfx950):", [round(x) for x in v[-4:]], "-> MB", round(2 * v[-1] / 1024, 2))
PY
done; done
