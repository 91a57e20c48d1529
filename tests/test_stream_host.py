"""The C++ host decoder's planning step (xvc_gpu::PictureDecoder::Plan,
xvc_amd/host/xvc_picture_decoder.cc), CPU only: the neighbour state it derives
for every intra CU of real streams must equal what the reference's
IntraPrediction::DetermineNeighbors saw (captured in the stream fixtures), and
its dependency waves must respect coding order dependencies."""
import numpy as np
import pytest

import stream_fixture as sf
from xvc_amd import decoder


@pytest.mark.parametrize("name", ["tiny", "c0", "c1"])
def test_plan_neighbors_equal_reference(name):
    fx = sf.StreamFixture(name)
    for i in range(fx.n):
        info, cus = fx.info[i], fx.cus(i)
        ps, cs = sf.to_syntax(info, cus)
        n_waves, nb, wave = decoder.plan_picture(ps, cs, fx.levels(i))
        assert n_waves >= 1
        intra = cus["pred_mode"] == 0
        for c in range(3):
            if info["two_trees"]:
                sel = intra & (cus["tree"] == (1 if c else 0))
            else:
                sel = intra
            assert np.array_equal(nb[sel, c], cus["nb_flags"][sel, c]), (name, i, c)
            assert np.array_equal(nb[sel, 3 + c], cus["nb_above_right"][sel, c]), (name, i, c)
            assert np.array_equal(nb[sel, 6 + c], cus["nb_below_left"][sel, c]), (name, i, c)
        # plain inter CUs need nobody: wave 0
        plain = (cus["pred_mode"] == 1) & (cus["lic"] == 0)
        assert np.all(wave[plain] == 0)
        assert wave.max() == n_waves - 1


def test_syntax_struct_sizes():
    assert sf.CU_SYNTAX_DTYPE == decoder.CU_SYNTAX_DTYPE
    assert sf.PICTURE_SYNTAX_DTYPE == decoder.PICTURE_SYNTAX_DTYPE
