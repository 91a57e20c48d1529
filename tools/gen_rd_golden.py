#!/usr/bin/env python3
"""The REST of a real encoder run's RD search (authoring container only: needs
oracle/_ref/libxvcref.so).

    python tools/gen_rd_golden.py [tiny] [c0] [c1]

tools/gen_me_golden.py records the uni-directional motion searches of the
reference encoder's RD search; this records what else the search calls on the
hot path while it codes the same clips (hooks in oracle/ref_harness.cc, no
control flow restated - the encoder runs its own code):

  steps   every bi-prediction refinement step (SearchBiIterative ->
          SearchRefIdx -> MotionEstNormal(kFullSearch): FullSearch + sub-pel on
          the 2 * orig - other-list target, inter_search.cc:392-433, :606-662)
          and every affine motion search (MotionEstAffine, :664-749, uni and
          bi): CU, lists / pictures, the two predictors, start predictor,
          bootstrap vector, the other list's vector(s), lambda -> vector(s),
          distortion, final predictor index;
  merges  every SearchMergeCandidates (:165-197): the five candidates ->
          the sorted order, the double costs, the count;
  calls   every TransformAndReconstruct of an inter CU
          (transform_encoder.cc:203-285): component, transform choice, scan ->
          QuantRdo's count, CRC-32 of the levels, CRC-32 of the reconstruction
          block, the returned distortion;
  evals   the CU state those calls share (motion, flags, qp) with the index of
          the CABAC context snapshot RdoQuant::QuantRdo read
          (rdo_quant.cc:254) and CompressAndEvalTransform's cbf-zero distortions;
  contexts / qps  the distinct snapshots (xvcgpu_rdoq_contexts) and the
          per-Qp fixed-point inputs (lambda, rd_factor, distortion weights);
  neighbours / nb_samples  for CUs that use local illumination compensation:
          the CUs above / left and the CURRENT reconstruction's row above / column
          left of the block at that moment of the RD search (what DeriveLicParams
          reads, inter_prediction.cc:1577-1663) - the only inputs that are not in
          the stream fixture's final pictures.

The bitstream must equal the committed stream fixture, so every reference
picture a call names is one of that fixture's reconstructions.  Written to
tests/golden/rd_calls_<clip>.npz (data only)."""
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
import rd_fixture  # noqa: E402
from rd_fixture import (STEP_DTYPE, MERGE_DTYPE, EVAL_DTYPE, QP_DTYPE, CALL_DTYPE,  # noqa: E402
                        NB_DTYPE, CTX_BYTES)

# clip -> the POC whose calls are kept (None: all pictures)
KEEP = {"tiny": None, "c0": None, "c1": 2, "c0q22": None, "c0q37": None}
DTYPES = [STEP_DTYPE, MERGE_DTYPE, EVAL_DTYPE, QP_DTYPE, CALL_DTYPE, None, NB_DTYPE,
          np.dtype("<u2")]
NAMES = ["steps", "merges", "evals", "qps", "calls", "contexts", "neighbours", "nb_samples"]


def fetch(lib, which):
    n = lib.xr_rd_count(which)
    size = lib.xr_rd_size(which)
    if which == 5:
        assert size == CTX_BYTES, size
        dt = np.dtype((np.uint8, CTX_BYTES))
    else:
        dt = DTYPES[which]
        assert size == dt.itemsize, (NAMES[which], size, dt.itemsize)
    if n == 0:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * size)).from_address(lib.xr_rd_data(which))
    return np.frombuffer(buf, dt).copy()


def main():
    lib = C.CDLL(ol.REF_SO)
    lib.xr_rd_count.restype = C.c_long
    lib.xr_rd_data.restype = C.c_void_p
    for name, only in KEEP.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        c = gsg.CLIPS[name]
        clip = synth.SyntheticClip(c["w"], c["h"], 8)
        lib.xr_rd_capture_begin(-1 if only is None else only)
        stream = gsg.encode(lib, clip, c["w"], c["h"], c["n"], c["qp"], c["sub_gop"], threads=0)
        lib.xr_rd_capture_end()
        committed = np.load(os.path.join(sf.GOLDEN, "stream_%s.npz" % name))["stream"]
        assert np.array_equal(stream, committed), "stream differs from the committed fixture"
        arrays = {NAMES[k]: fetch(lib, k) for k in range(8)}
        calls, evals, steps = arrays["calls"], arrays["evals"], arrays["steps"]
        print("  %s: %d bi / affine steps (bi %d, affine uni %d, affine bi %d; LIC CUs %d), "
              "%d merge rankings, %d transform calls (%d completed) in %d CU states "
              "(LIC %d, affine %d, bi %d), %d context snapshots, %d qps, %d LIC neighbourhoods "
              "(%d samples); %d intra-CU transform calls not kept" % (
                  name, len(steps), int((steps["kind"] == 1).sum()),
                  int((steps["kind"] == 2).sum()), int((steps["kind"] == 3).sum()),
                  int(((steps["flags"] & 2) != 0).sum()), len(arrays["merges"]), len(calls),
                  int(calls["completed"].sum()), len(evals), int(((evals["flags"] & 2) != 0).sum()),
                  int(((evals["flags"] & 8) != 0).sum()), int((evals["inter_dir"] == 2).sum()),
                  len(arrays["contexts"]), len(arrays["qps"]), len(arrays["neighbours"]),
                  len(arrays["nb_samples"]), lib.xr_rd_count(8)))
        path = os.path.join(sf.GOLDEN, "rd_calls_%s.npz" % name)
        np.savez_compressed(path, **rd_fixture.to_columns(arrays))
        print("  -> %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))
        gsg.update_manifest("rd_calls_%s.npz" % name)


if __name__ == "__main__":
    main()
