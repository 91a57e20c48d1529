#!/usr/bin/env python3
"""tests/golden/cs_builder_digests.json: SHA-256 of every array and of a set of programs
xvc_gpu::CuStateBuilder produces for the captured pictures.  Taken in round 6 while the Python
composer of rounds 4 - 5 (tests/rd_serial.py build_passes / build_merge_folds / program) still
existed and both agreed byte for byte on tiny, c0 (POC 2, 4) and c1 (/tmp check script of that
round); since then the digests pin the C++ composer.

    python tools/gen_cs_builder_digests.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_cu_state_builder as t  # noqa: E402

out = {}
for name, poc in (("tiny", 2), ("c0", 2), ("c0", 4), ("c1", 2)):
    d, sp, b = t.digests(name, poc)
    out["%s_%d" % (name, poc)] = d
    b.destroy()
    print(name, poc, len(d))
json.dump(out, open(t.GOLDEN, "w"), indent=1, sort_keys=True)
print("->", t.GOLDEN)
