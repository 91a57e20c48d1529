#!/usr/bin/env python3
"""How much of a picture's RD search could be in flight at once: the dependency DAG of its
CU states (tests/rd_serial.py: the reference's issue order) and its critical path.

What a state READS that an earlier state of the same picture DECIDED
(cu_encoder.cc:123-273 CompressCu, :431-541 CompressInterPic):
  * the motion / mode / reconstruction of the CUs left of, above and diagonal to it (AMVP
    and merge candidates, inter_prediction.cc GetMvpList / GetMergeCandidates; intra
    reference samples and LIC neighbours; the deblocking-free reconstruction) - every
    earlier state whose CU covers a sample of the one-sample ring around this CU
    (left column, above row, the corners, above-right and below-left extensions);
  * a merge candidate's evaluation reads its CU's merge ranking.
What it does NOT read as data: the other modes of its own CU, its parent's modes (the
reference orders those only for best_cu_cost pruning and CuCache hints) - and, NOT MODELLED
here, the CABAC context state the bit costs are priced with, which the reference restores
per CU from the state behind everything coded before it (cu_encoder.cc:166-170): with
that chain every state depends on its predecessor and the parallelism is 1.  The figure
below is therefore an UPPER bound on what a bit-exact device walk could overlap, the bound
a walk that is handed the context snapshots (as the captured walk is) can approach.
Regions are tracked on a 4x4-sample grid, each cell holding the longest chain that ends in
a state covering it (a parent's states count towards its children's neighbours: that can
only lengthen the path).

    python tools/state_dag.py [tiny|c0|c1] [poc]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = 4   # grid granularity in luma samples


def critical_path(states, weights=None):
    """states: the xvc_cs_state table (issue order).  weights: cost per state kind (None:
    every state counts 1).  Returns dict(states, critical_path, parallelism, ...)."""
    st = states[states["supported"] != 0]
    n = len(st)
    if n == 0:
        return None
    x0, y0 = st["x"].astype(np.int64), st["y"].astype(np.int64)
    w, h = st["w"].astype(np.int64), st["h"].astype(np.int64)
    W = int((x0 + w).max()) + 2 * 64
    H = int((y0 + h).max()) + 2 * 64
    off = 64 // G                      # a margin so that rings never leave the grid
    grid = np.zeros((H // G + 2 * off, W // G + 2 * off), np.float64)
    kind = st["kind"].astype(np.int64)
    cost = np.ones(n) if weights is None else np.array([weights[int(k)] for k in kind], np.float64)
    depth = np.zeros(n)
    last_rank = {}
    for i in range(n):
        cx, cy = int(x0[i]) // G + off, int(y0[i]) // G + off
        cw, ch = max(int(w[i]) // G, 1), max(int(h[i]) // G, 1)
        # the ring: above row with its corners and the above-right extension, left column
        # with the below-left extension
        d = max(grid[cy - 1, cx - 1:cx + 2 * cw].max(), grid[cy:cy + 2 * ch, cx - 1].max())
        key = (int(x0[i]), int(y0[i]), int(w[i]), int(h[i]))
        if kind[i] == 0:
            last_rank[key] = i
        elif kind[i] == 1 and key in last_rank:
            d = max(d, depth[last_rank[key]])
        depth[i] = d + cost[i]
        cell = grid[cy:cy + ch, cx:cx + cw]
        np.maximum(cell, depth[i], out=cell)
    total = float(cost.sum())
    cp = float(depth.max())
    return {"states": int(n), "work": round(total, 3), "critical_path": round(cp, 3),
            "parallelism": round(total / cp, 2),
            "weights": "1 per state" if weights is None else "cost per state kind"}


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import rd_serial
    from xvc_amd import api
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    poc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sp = rd_serial.SerialPicture(api, name, poc)
    print(json.dumps({"clip": name, "poc": poc, "by_count": critical_path(sp.states)}, indent=1))


if __name__ == "__main__":
    main()
