#!/usr/bin/env python3
"""xvc_amd/data/rdoq_init_contexts.npy: the coefficient-coding CABAC context
states a syntax writer starts a picture with, for every picture qp 0..63 and
picture type (0 bi, 1 uni, 2 intra) - [64][3] records of xvcgpu_rdoq_contexts
(152 bytes).  RDOQ reads the entropy coder's contexts (rdo_quant.cc:254); a host
that runs the real entropy coder snapshots its live states per batch, the frame
pass of this repo (no entropy coding) feeds the picture-initial ones.  Captured
from the reference build (CabacContexts::ResetStates through
oracle/_ref: xr_rdoq_init_contexts); authoring container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import oracle_rdoq as oq  # noqa: E402


def table(xr, bd=10):
    out = np.zeros((64, 3), oq.RDOQ_CTX_DTYPE)
    for qp in range(64):
        for t in range(3):
            out[qp, t] = oq.init_contexts(xr, bd, qp, t)[0]
    return out


if __name__ == "__main__":
    xr = ol.Lib("xr")
    t = table(xr)
    assert all(np.array_equal(t, table(xr, bd)) for bd in (8, 12))   # bit depth plays no part
    path = os.path.join(ROOT, "xvc_amd", "data", "rdoq_init_contexts.npy")
    np.save(path, t.view(np.uint8).reshape(64, 3, -1))
    print(path, os.path.getsize(path), "bytes")
