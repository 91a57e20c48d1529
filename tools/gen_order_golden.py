#!/usr/bin/env python3
"""The ORDER of a real encoder run's RD search, and the motion searches' prices
(authoring container only: needs oracle/_ref/libxvcref.so).

    python tools/gen_order_golden.py [tiny] [c0] [c1]

tools/gen_me_golden.py and tools/gen_rd_golden.py record WHAT the reference
encoder's RD search computes on the hot path - each table in its own capture
order.  This run records both in one encode and adds

  seq/<table>  a sequence number over ALL tables (me calls, steps, merges, evals,
          calls, cands, finals): the order in which CuEncoder::CompressCu
          (cu_encoder.cc:123-273) really issued the work - one CU state after
          the other, each a chain whose results feed the next;
  cands   every candidate InterSearch::SearchRefIdx prices
          (inter_search.cc:556-571): list, picture, vector, final predictor,
          distortion and the bits GetInterPredBits returned for it with the
          encoder's DEFAULT setting (fast_inter_pred_bits == 0: a throw-away
          RdoSyntaxWriter on the live CABAC state, :1131-1135);
  ictx    the states of the contexts CuWriter::WriteInterPrediction reads for
          that CU (include/xvcgpu_types.h: xvcgpu_inter_contexts);
  finals  the motion state SearchMotion ends with (:247-257).

The stream must equal the committed fixture and the me / rd tables of this run
must equal the committed me_calls / rd_calls fixtures record for record: the
sequence numbers index THOSE.  Written to tests/golden/rd_order_<clip>.npz."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_me_golden as gmg  # noqa: E402
import gen_rd_golden as grg  # noqa: E402
import gen_stream_golden as gsg  # noqa: E402
import oracle_lib as ol  # noqa: E402
import order_fixture as of  # noqa: E402
import rd_fixture as rf  # noqa: E402
import stream_fixture as sf  # noqa: E402
from xvc_amd import synth  # noqa: E402

KEEP = {"tiny": None, "c0": None, "c1": 2}
SEQ_TABLES = ["me", "steps", "merges", "evals", "calls", "cands", "finals"]


def fetch(lib, which, dt):
    n = lib.xr_rd_count(which)
    assert lib.xr_rd_size(which) == dt.itemsize, (which, lib.xr_rd_size(which), dt.itemsize)
    if n == 0:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(lib.xr_rd_data(which))
    return np.frombuffer(buf, dt).copy()


def main():
    lib = C.CDLL(ol.REF_SO)
    lib.xr_rd_count.restype = C.c_long
    lib.xr_rd_data.restype = C.c_void_p
    lib.xr_me_capture_end.restype = C.c_long
    lib.xr_me_calls.restype = C.c_void_p
    for name, only in KEEP.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        c = gsg.CLIPS[name]
        clip = synth.SyntheticClip(c["w"], c["h"], 8)
        lib.xr_me_capture_begin(-1 if only is None else only)
        lib.xr_rd_capture_begin(-1 if only is None else only)
        stream = gsg.encode(lib, clip, c["w"], c["h"], c["n"], c["qp"], c["sub_gop"], threads=0)
        n_me = lib.xr_me_capture_end()
        lib.xr_rd_capture_end()
        committed = np.load(os.path.join(sf.GOLDEN, "stream_%s.npz" % name))["stream"]
        assert np.array_equal(stream, committed), "stream differs from the committed fixture"
        # the tables of this run = the committed fixtures
        buf = (C.c_char * (n_me * gmg.CALL_DTYPE.itemsize)).from_address(lib.xr_me_calls())
        me = np.frombuffer(buf, gmg.CALL_DTYPE)
        assert np.array_equal(me, np.load(os.path.join(sf.GOLDEN, "me_calls_%s.npz" % name))["calls"])
        rd = rf.load(name)
        for k, t in enumerate(grg.NAMES[:5]):
            got = grg.fetch(lib, k)
            want = rd[t]
            if t == "calls":       # the fixture keeps 16 bits of the reconstruction's CRC
                got = got.copy()
                got["rec_crc"] &= 0xffff
            assert len(got) == len(want) and all(
                np.array_equal(got[f], want[f]) for f in want.dtype.names if not f.startswith("pad")), t
        out = {"cands": fetch(lib, 11, of.CAND_DTYPE), "finals": fetch(lib, 12, of.FINAL_DTYPE),
               "ictx": fetch(lib, 13, of.ICTX_DTYPE)}
        seqs = [fetch(lib, 20 + t, np.dtype("<u4")) for t in range(7)]
        sizes = [n_me, len(rd["steps"]), len(rd["merges"]), len(rd["evals"]), len(rd["calls"]),
                 len(out["cands"]), len(out["finals"])]
        assert [len(s) for s in seqs] == sizes, ([len(s) for s in seqs], sizes)
        allseq = np.sort(np.concatenate(seqs))
        assert np.array_equal(allseq, np.arange(len(allseq), dtype=np.uint32) + allseq[0])
        cols = of.to_columns(out)
        for t, s in zip(SEQ_TABLES, seqs):
            cols["seq/" + t] = np.diff(s.astype(np.int64) - int(allseq[0]), prepend=0).astype(np.int32)
        path = os.path.join(sf.GOLDEN, "rd_order_%s.npz" % name)
        np.savez_compressed(path, **cols)
        cd = out["cands"]
        print("  %s: %d records in order; %d priced candidates (uni %d, re-used %d, bi %d, affine "
              "uni %d, affine bi %d), %d context snapshots, %d SearchMotion results -> %s (%.1f KB)"
              % (name, len(allseq), len(cd), int(((cd["kind"] == 0) & (cd["reused"] == 0)).sum()),
                 int(cd["reused"].sum()), int((cd["kind"] == 1).sum()), int((cd["kind"] == 2).sum()),
                 int((cd["kind"] == 3).sum()), len(out["ictx"]), len(out["finals"]), path,
                 os.path.getsize(path) / 1024))
        gsg.update_manifest("rd_order_%s.npz" % name)


if __name__ == "__main__":
    main()
