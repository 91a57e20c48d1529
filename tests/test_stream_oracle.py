"""The oracle's decoder reconstruction (oracle/xvc_oracle_dec.c) pinned against
the reference decoder on real streams: tests/golden/stream_*.npz hold the parsed
syntax and the output of the reference decoder for streams the reference
encoder produced (tools/gen_stream_golden.py).  CPU only."""
import numpy as np
import pytest

import stream_fixture as sf


def _check_fixture(name, max_pics=None):
    fx = sf.StreamFixture(name)
    n = fx.n if max_pics is None else min(fx.n, max_pics)
    pics = [(fx.info[i], fx.cus(i), fx.levels(i)) for i in range(n)]

    def check(i, pic, pre, nb):
        info, cus = fx.info[i], fx.cus(i)
        intra = cus["pred_mode"] == 0
        # the neighbour state every intra CU saw == IntraPrediction::DetermineNeighbors
        for c in range(3):
            comps = intra & ((cus["tree"] == 1) if c and info["two_trees"] else
                             (cus["tree"] == 0))
            if c and info["two_trees"]:
                comps = intra & (cus["tree"] == 1)
            elif info["two_trees"]:
                comps = intra & (cus["tree"] == 0)
            assert np.array_equal(nb[comps, c], cus["nb_flags"][comps, c]), (name, i, c)
            assert np.array_equal(nb[comps, 3 + c], cus["nb_above_right"][comps, c]), (name, i, c)
            assert np.array_equal(nb[comps, 6 + c], cus["nb_below_left"][comps, c]), (name, i, c)
        if fx.has_planes(i, "pre"):
            for c, e in enumerate(fx.planes(i, "pre")):
                assert np.array_equal(pre[c], e), "%s pic %d comp %d before the filter" % (name, i, c)
        if fx.has_planes(i, "post"):
            for c, e in enumerate(fx.planes(i, "post")):
                assert np.array_equal(pic.planes[c], e), "%s pic %d comp %d" % (name, i, c)
        md5 = sf.picture_md5(pic.planes, int(info["bitdepth"]))
        assert np.array_equal(md5, info["md5"]), "%s pic %d MD5" % (name, i)

    sf.oracle_decode_stream(pics, check)


def test_oracle_decodes_tiny_stream():
    _check_fixture("tiny")


def test_oracle_decodes_cif_stream():
    """BASELINE config 0: CIF, 10 frames, QP 32, xvcenc defaults."""
    _check_fixture("c0")


def test_oracle_decodes_1080p_stream():
    """BASELINE config 1 content (1080p QP 32): every picture's MD5."""
    _check_fixture("c1")
