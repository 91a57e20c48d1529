#!/usr/bin/env python3
"""The reference encoder's own time on the clip the RD-search walk takes its picture from
(c1: 1920x1080, 5 frames, QP 32, sub-GOP 4 - tools/gen_stream_golden.py), through
oracle/_ref/libxvcref.so (the reference built by oracle/Makefile; the library travels to the
GPU box, /root/reference does not), on this host's cores: one thread and the encoder's own
thread pool (threads = -1: xvcenc's default, hardware_concurrency; thread_encoder.cc).  The CPU figure that
stands beside bench.py's encoder_rd_serial (which quotes the newest committed
profiles/rNN_ref_encoder_cpu.json).

    python tools/ref_encoder_time.py [--frames 5] > profiles/r06_ref_encoder_cpu.json"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import oracle_lib as ol  # noqa: E402
import stream_fixture as sf  # noqa: E402
from xvc_amd import synth  # noqa: E402

W, H, QP, SUB_GOP = 1920, 1080, 32, 4


def encode(lib, frames, n, threads):
    cap = 64 << 20
    out = np.zeros(cap, np.uint8)
    lib.xr_stream_encode.restype = C.c_long
    lib.xr_stream_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_long]
    t = time.perf_counter()
    used = lib.xr_stream_encode(W, H, 8, 0, 30.0, QP, SUB_GOP, -1, -1, threads, None, n,
                                frames.ctypes.data, out.ctypes.data, cap)
    dt = time.perf_counter() - t
    assert used > 0, used
    return dt, out[:used].copy()


def main():
    n = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 5
    lib = C.CDLL(ol.REF_SO)
    clip = synth.SyntheticClip(W, H, 8)
    frames = np.concatenate([np.concatenate([p.reshape(-1) for p in clip.frame(i)])
                             for i in range(n)]).astype(np.uint8)
    res = {"workload": "reference encoder (xvc_encoder_api through oracle/_ref/libxvcref.so), "
                       "%dx%d, %d frames of the synthetic clip, QP %d, sub-GOP %d: the clip of "
                       "tests/golden/stream_c1.npz whose POC 2 bench.py's encoder_rd_serial walks" %
                       (W, H, n, QP, SUB_GOP),
           "host_cores": os.cpu_count(), "frames": n, "runs": {}}
    ref = None
    for name, threads in (("threads_1", 1), ("threads_auto", -1)):
        dt, stream = encode(lib, frames, n, threads)
        if ref is None:
            ref = stream
        res["runs"][name] = {"threads": threads if threads > 0 else "-1: the encoder's own choice (hardware_concurrency = %d)" % os.cpu_count(),
                             "seconds": round(dt, 2), "pictures_per_s": round(n / dt, 4),
                             "inter_pictures_per_s_upper_bound": round((n - 1) / dt, 4),
                             "stream_equal_to_first_run": bool(np.array_equal(stream, ref))}
    path = os.path.join(sf.GOLDEN, "stream_c1.npz")
    if n == 5 and os.path.exists(path):
        res["stream_equals_committed_fixture"] = bool(np.array_equal(np.load(path)["stream"], ref))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
