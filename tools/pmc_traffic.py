"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into
profiles/traffic.json: average HBM bytes per launch for every kernel.

FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B?  rocprofv3 reports them
in kilobytes; on gfx950 FETCH_SIZE of wide coalesced reads reports half the
bytes (MI355X_MICROARCH.md, HBM section) -> the read side is doubled, as the
guide prescribes; WRITE_SIZE is taken as reported (uncalibrated).

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel_source_md5 quant]
(with the last two the file is the profiles/traffic_current.json bench.py reads:
it carries the MD5 of the kernel sources and the quantiser it was taken with)
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kb, w_kb = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"fetch_kb_reported": f_kb, "write_kb_reported": w_kb,
              "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
              "correction": "FETCH_SIZE x2 (gfx950 half-count of wide reads), WRITE_SIZE as is"}
if len(sys.argv) > 5:
    out = {"kernel_source_md5": sys.argv[4], "quant": sys.argv[5],
           "command": "python bench.py --no-cpu --no-decode --quant %s --steps 56 --warmup 14" % sys.argv[5],
           "kernels": out}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    out = out["kernels"]
else:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 3) for k, v in out.items()}))
