#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of bench.py and prints, for the second
half of the run: the span, the time at least one kernel was running, the
average number of kernels running, and per kernel its share of the span.
usage: timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0],
             r.get("Queue_Id", "")) for r in rows)
t0, t1 = ev[len(ev) // 2][0], ev[-1][1]
ev = [e for e in ev if e[0] >= t0]
span = t1 - t0
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
busy = conc = 0
depth = 0
last = t0
hist = collections.Counter()
for t, d in pts:
    if depth > 0:
        busy += t - last
    conc += depth * (t - last)
    hist[depth] += t - last
    depth += d
    last = t
print("span %.2f ms, some kernel running %.1f%%, mean kernels in flight %.2f" %
      (span / 1e6, 100.0 * busy / span, conc / span))
print("time by number of kernels running:", {k: "%.1f%%" % (100.0 * v / span) for k, v in sorted(hist.items())})
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in ev:
    per[n][0] += e - s
    per[n][1] += 1
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("  %-50s %5d launches, avg %8.1f us, %5.1f%% of span" % (n[:50], c, d / c / 1e3, 100.0 * d / span))
print("queues:", collections.Counter(e[3] for e in ev))
