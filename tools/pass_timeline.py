#!/usr/bin/env python3
"""What the frame passes of bench.py's chains look like on the device's time line.

    rocprofv3 --kernel-trace -d <dir> -o tl -- python bench.py --no-cpu --no-decode --steps 300
    python tools/pass_timeline.py <dir>/tl_results.db

From the kernel dispatches of the middle third of the trace (the timed region):
per kernel its average duration WITH the other chains in flight, the idle time of
a queue between two kernels of one chain, how many kernels are in flight at a time,
and the time per frame pass.  A chain is latency bound (each kernel waits for the
one before it); the chains together fill the chip - the sum of the in-flight
durations divided by the number of chains is the pass time one sees."""
import collections
import sqlite3
import sys

import numpy as np


def short(name):
    name = name.split("(")[0]
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:44]


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    name_col = "display_name" if "display_name" in \
        [r[1] for r in db.execute("pragma table_info(%s)" % ks)] else "kernel_name"
    rows = list(db.execute(
        "select d.start, d.end, d.queue_id, s.%s from %s d join %s s on d.kernel_id = s.id "
        "order by d.start" % (name_col, kd, ks)))
    n = len(rows)
    rows = rows[n // 3:2 * n // 3]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    print("%d dispatches on %d queues (middle third of the trace)" % (len(rows), len(byq)))
    dur = collections.defaultdict(list)
    for r in rows:
        dur[short(r[3])].append((r[1] - r[0]) / 1e3)
    first = None
    print("%-46s %7s %9s %9s %9s" % ("kernel", "calls", "mean us", "min us", "p90 us"))
    total = 0.0
    per_pass = max(len(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if len(v) * 4 < per_pass:
            continue    # blits etc.
        print("%-46s %7d %9.1f %9.1f %9.1f" % (k, len(v), np.mean(v), np.min(v),
                                                np.percentile(v, 90)))
        total += np.mean(v) * len(v) / per_pass
    gaps = collections.defaultdict(list)
    for q, v in byq.items():
        for a, b in zip(v[:-1], v[1:]):
            gaps[(short(a[3]), short(b[3]))].append((b[0] - a[1]) / 1e3)
    print("\nqueue idle between consecutive kernels of a chain (mean / median us):")
    gap_total = 0.0
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) * 4 < per_pass:
            continue
        print("  %-36s -> %-36s %7.1f %7.1f" % (k[0][:36], k[1][:36], np.mean(v), np.median(v)))
        gap_total += np.mean(v) * len(v) / per_pass
    ev = []
    for r in rows:
        ev.append((r[0], 1))
        ev.append((r[1], -1))
    ev.sort()
    cur, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    tot = sum(hist.values())
    print("\nkernels in flight (share of the time): " +
          ", ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e3
    print("sum of a pass's in-flight kernel durations %.1f us + queue idle %.1f us = %.1f us per "
          "chain and pass; %d chains -> %.1f us per pass expected, %.1f us measured over the "
          "window" % (total, gap_total, total + gap_total, len(byq),
                      (total + gap_total) / len(byq), span / per_pass))


if __name__ == "__main__":
    main(sys.argv[1])
