"""Summarises the SQ counter passes of tools/profile_bench.sh into the file
bench.py reads for its `roofline.valu_issue` entry (profiles/issue_current.json):
wave instructions per launch of every kernel of the pass - vector, scalar, LDS -,
waves, active lanes and LDS bank conflicts, averaged over the LAST fifth of the
launches (the chain's settled state), stamped with the MD5 of the kernel sources.

usage: pmc_issue.py <profile_dir with sq*/> <out.json> <kernel_source_md5> <quant>"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/sq*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    tail = {c: (sum(v[-max(1, len(v) // 5):]) / max(1, len(v) // 5)) for c, v in cs.items()}
    if tail.get("SQ_INSTS_VALU", 0) < 1000:
        continue
    valu = tail["SQ_INSTS_VALU"]
    out[k] = {"valu": valu, "salu": tail.get("SQ_INSTS_SALU"), "lds": tail.get("SQ_INSTS_LDS"),
              "waves": tail.get("SQ_WAVES"),
              "active_lanes": (tail["SQ_THREAD_CYCLES_VALU"] / 64.0 / valu
                               if "SQ_THREAD_CYCLES_VALU" in tail else None),
              "lds_bank_conflict_share": (tail["SQ_LDS_BANK_CONFLICT"] / tail["SQ_LDS_IDX_ACTIVE"]
                                          if tail.get("SQ_LDS_IDX_ACTIVE") else None)}
json.dump({"kernel_source_md5": sys.argv[3], "quant": sys.argv[4],
           "counters": "SQ_INSTS_VALU / _SALU / _LDS, SQ_WAVES, SQ_THREAD_CYCLES_VALU, "
                       "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch, mean of the last fifth "
                       "of the launches of tools/profile_bench.sh's SQ passes",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["valu"] / 1e6, 2) for k, v in out.items()}))
