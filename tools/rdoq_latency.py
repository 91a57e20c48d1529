#!/usr/bin/env python3
"""Latency of the RDOQ walk for ONE block (run on the GPU box): the serial
chain that bounds quant_rdo_packed_kernel.  Dense coefficients = every
sub-block live = the longest walk a block of that size can take."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from xvc_amd import api, pipeline  # noqa: E402

ctx = api.Context(0)
rng = np.random.default_rng(1)
bd, qp = 10, 32
ctxs = pipeline.rdoq_init_contexts(qp, 1)
lam, rdf = pipeline.rdoq_host_params(qp, bd)[0]
for (w, h, n, dense) in [(16, 16, 1, 1), (16, 16, 1, 0), (8, 8, 1, 1), (32, 32, 1, 1),
                         (16, 16, 4096, 1), (16, 16, 4096, 0)]:
    blocks = np.zeros(n, api.TX_DTYPE)
    blocks["w"], blocks["h"], blocks["qp"], blocks["intra_pic"] = w, h, qp, api.TXF_RDOQ
    prm = np.zeros(n, api.RDOQ_PARAMS_DTYPE)
    prm["lambda"], prm["rd_factor"] = lam, rdf
    amp = 400.0 if dense else 60.0
    yy, xx = np.mgrid[0:h, 0:w]
    decay = 1.0 if dense else np.exp(-(xx + yy) / 2.0)
    cf = np.clip(np.rint(rng.laplace(0, 1, (n, h, w)) * amp * decay), -32768, 32767).astype(np.int16)
    off = (np.arange(n) * w * h).astype(np.uint32)
    db, dof, dcf = ctx.buffer(blocks), ctx.buffer(off), ctx.buffer(cf.reshape(-1))
    dc, dp = ctx.buffer(ctxs), ctx.buffer(prm)
    dl, dn = ctx.alloc(2 * cf.size), ctx.alloc(4 * n)

    def run():
        ctx._check(ctx.lib.xvcgpu_quant_rdo_batch(ctx.h, bd, db.ptr, n, dcf.ptr, dof.ptr, cf.size,
                                                  dl.ptr, dn.ptr, dc.ptr, dp.ptr))
    run()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(10):
        run()
    ms = ctx.timer_end() / 10
    nnz = dn.to_array(np.int32, n)
    print("%dx%d x %d %s: %.1f us per batch, mean nnz %.1f" %
          (w, h, n, "dense" if dense else "sparse", 1e3 * ms, nnz.mean()))
