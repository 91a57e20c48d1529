#!/usr/bin/env python3
"""The intra search's SATD pre-selection of a real encoder run (authoring container
only: needs oracle/_ref/libxvcref.so).

    python tools/gen_intra_golden.py [tiny] [c0]

While the reference encoder codes a clip of tools/gen_stream_golden.py, hooks in
oracle/ref_harness.cc (xr_intra) record a sample of its
IntraSearch::DetermineSlowIntraModes calls (intra_search.cc:188-305): the CU, what
DetermineNeighbors said about its surroundings, the CURRENT reconstruction's row
above and column to the left (the reference samples of that moment of the RD
search - not in any final picture) and, for every mode it evaluated, the SATD.
Written to tests/golden/intra_calls_<clip>.npz (data only); replayed through
xvcgpu_intra_satd_batch by tests/test_gpu_intra_calls.py."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_stream_golden as gsg  # noqa: E402
import oracle_lib as ol  # noqa: E402
import stream_fixture as sf  # noqa: E402
from xvc_amd import synth  # noqa: E402
from intra_fixture import CALL_DTYPE, EVAL_DTYPE, ITX_DTYPE  # noqa: E402
from rd_fixture import QP_DTYPE, CTX_BYTES  # noqa: E402

# clip -> (calls kept at most, every n-th call)
KEEP = {"tiny": (1500, 3), "c0": (3000, 11), "c1": (3000, 199), "c0q22": (1200, 17),
        "c0q37": (1200, 13)}
# clip -> (TransformAndReconstruct calls of intra CUs kept at most, every n-th, only this POC)
KEEP_TX = {"tiny": (4000, 53, -1), "c0": (6000, 397, -1), "c1": (6000, 2003, 0),
           "c0q22": (3000, 397, -1), "c0q37": (3000, 307, -1)}


def fetch(lib, which, dt):
    n = lib.xr_intra_count(which)
    assert lib.xr_intra_size(which) == dt.itemsize, (which, lib.xr_intra_size(which), dt.itemsize)
    if n == 0:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(lib.xr_intra_data(which))
    return np.frombuffer(buf, dt).copy()


def main():
    lib = C.CDLL(ol.REF_SO)
    lib.xr_intra_count.restype = C.c_long
    lib.xr_intra_data.restype = C.c_void_p
    lib.xr_rd_count.restype = C.c_long
    lib.xr_rd_data.restype = C.c_void_p
    for name, (cap, stride) in KEEP.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        c = gsg.CLIPS[name]
        clip = synth.SyntheticClip(c["w"], c["h"], 8)
        lib.xr_intra_capture_begin(cap, stride)
        tcap, tstride, tpoc = KEEP_TX[name]
        lib.xr_rd_capture_begin(tpoc)        # (the inter half of that capture is not kept here)
        lib.xr_rd_capture_intra(tstride, tcap)
        stream = gsg.encode(lib, clip, c["w"], c["h"], c["n"], c["qp"], c["sub_gop"], threads=0)
        lib.xr_intra_capture_end()
        lib.xr_rd_capture_end()
        committed = np.load(os.path.join(sf.GOLDEN, "stream_%s.npz" % name))["stream"]
        assert np.array_equal(stream, committed), "stream differs from the committed fixture"
        calls = fetch(lib, 0, CALL_DTYPE)
        evals = fetch(lib, 1, EVAL_DTYPE)
        samples = fetch(lib, 2, np.dtype("<u2"))
        sizes = sorted(set(zip(calls["w"].tolist(), calls["h"].tolist())))
        print("  %s: %d of %d DetermineSlowIntraModes calls kept (pictures %s), %d mode evaluations, "
              "%d neighbour samples; sizes %s" % (name, len(calls), lib.xr_intra_count(3),
                                                 sorted(set(calls["poc"].tolist())), len(evals),
                                                 len(samples), sizes))
        path = os.path.join(sf.GOLDEN, "intra_calls_%s.npz" % name)
        cols = {"calls/" + f: np.ascontiguousarray(calls[f]) for f in CALL_DTYPE.names
                if not f.startswith("pad")}
        cols.update({"evals/" + f: np.ascontiguousarray(evals[f]) for f in EVAL_DTYPE.names
                     if not f.startswith("pad")})
        cols["samples"] = samples
        # the intra CUs' TransformAndReconstruct sample, with the context snapshots and
        # per-Qp inputs it names (renumbered to the ones in use)
        def rd_fetch(which, dt):
            n = lib.xr_rd_count(which)
            assert lib.xr_rd_size(which) == dt.itemsize, (which, lib.xr_rd_size(which), dt.itemsize)
            if n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(lib.xr_rd_data(which))
            return np.frombuffer(buf, dt).copy()
        itx = rd_fetch(9, ITX_DTYPE)
        ctxs = rd_fetch(5, np.dtype((np.uint8, CTX_BYTES)))
        qps = rd_fetch(3, QP_DTYPE)
        uc, ic = np.unique(itx["ctx_index"], return_inverse=True)
        uq, iq = np.unique(itx["qp_index"], return_inverse=True)
        itx["ctx_index"], itx["qp_index"] = ic, iq
        cols.update({"itx/" + f: np.ascontiguousarray(itx[f]) for f in ITX_DTYPE.names})
        cols["itx_samples"] = rd_fetch(10, np.dtype("<u2"))
        cols["contexts"] = ctxs[uc]
        cols["qps"] = qps[uq].view(np.uint8).reshape(len(uq), -1)
        print("  %s: %d of %d TransformAndReconstruct calls of intra CUs kept (%d completed; comps %s; "
              "scans %s; DST 4x4 %d, transform skip %d, other types %d), %d context snapshots" % (
                  name, len(itx), lib.xr_rd_count(8), int(itx["completed"].sum()),
                  np.bincount(itx["comp"], minlength=3).tolist(),
                  np.bincount(itx["scan"], minlength=3).tolist(),
                  int(((itx["dst4x4"] != 0) & (itx["w"] == 4) & (itx["h"] == 4)).sum()),
                  int(itx["tx_skip"].sum()), int((itx["tx_hor"] > 1).sum()), len(uc)))
        np.savez_compressed(path, **cols)
        print("  -> %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))
        gsg.update_manifest("intra_calls_%s.npz" % name)


if __name__ == "__main__":
    main()
