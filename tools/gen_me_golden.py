#!/usr/bin/env python3
"""Motion-search calls of a real encoder run (authoring container only: needs
oracle/_ref/libxvcref.so).

    python tools/gen_me_golden.py

Re-encodes the clips of tools/gen_stream_golden.py with the REFERENCE encoder
(single-threaded) while the harness hook records every uni-directional motion
search the encoder's RD search makes (InterSearch::MotionEstNormal: TZ search
+ sub-pel refinement): block geometry, the AMVP predictor it was started from,
the previous CU's full-pel vector, lambda, search range, and the reference's
results.  The bitstream must come out identical to the committed stream
fixture, so the reference pictures of every call are the fixture's
reconstructions.  Written to tests/golden/me_calls_<clip>.npz (data only)."""
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

CALL_DTYPE = np.dtype([
    ("poc", "<i4"), ("ref_poc", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "u1"), ("h", "u1"),
    ("depth_nonzero", "u1"), ("fullpel_mv", "u1"), ("use_lic", "<i4"), ("mvp_x", "<i4"), ("mvp_y", "<i4"),
    ("prev_x", "<i4"), ("prev_y", "<i4"), ("lambda16", "<u4"), ("search_range", "<i4"),
    ("fullpel_x", "<i4"), ("fullpel_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4"),
    ("dist", "<u4")])

# clip -> POCs whose calls are kept (None: all pictures)
KEEP = {"tiny": None, "c1": [2], "c0": None, "c0q22": None, "c0q37": None}


def main():
    lib = C.CDLL(ol.REF_SO)
    assert lib.xr_me_call_size() == CALL_DTYPE.itemsize, lib.xr_me_call_size()
    lib.xr_me_capture_end.restype = C.c_long
    lib.xr_me_calls.restype = C.c_void_p
    for name, only in KEEP.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        c = gsg.CLIPS[name]
        clip = synth.SyntheticClip(c["w"], c["h"], 8)
        lib.xr_me_capture_begin(-1 if only is None else only[0])
        for extra in (only or [])[1:]:
            lib.xr_me_capture_also(extra)
        stream = gsg.encode(lib, clip, c["w"], c["h"], c["n"], c["qp"], c["sub_gop"], threads=0)
        n = lib.xr_me_capture_end()
        committed = np.load(os.path.join(sf.GOLDEN, "stream_%s.npz" % name))["stream"]
        assert np.array_equal(stream, committed), "stream differs from the committed fixture"
        buf = (C.c_char * (n * CALL_DTYPE.itemsize)).from_address(lib.xr_me_calls())
        calls = np.frombuffer(buf, CALL_DTYPE).copy()
        print("  %s: %d motion searches kept; block sizes: %s" % (
            name, n, sorted({(int(a), int(b)) for a, b in zip(calls["w"], calls["h"])})[:40]))
        np.savez_compressed(os.path.join(sf.GOLDEN, "me_calls_%s.npz" % name), calls=calls)
        gsg.update_manifest("me_calls_%s.npz" % name)


if __name__ == "__main__":
    main()
