"""The C++ host decoder's planning step (xvc_gpu::PictureDecoder::Plan,
xvc_amd/host/xvc_picture_decoder.cc), CPU only: the neighbour state it derives
for every intra CU of real streams must equal what the reference's
IntraPrediction::DetermineNeighbors saw (captured in the stream fixtures), and
its dependency waves must respect coding order dependencies."""
import numpy as np
import pytest

import stream_fixture as sf
from xvc_amd import decoder


@pytest.mark.parametrize("name", ["tiny", "c0", "c1", "c1x", "c0q22", "c0q37"])
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


def _mutations():
    """(name, function that breaks one field of the syntax) - each makes the
    picture something a bitstream could claim but the reconstruction must not
    be asked to index with."""
    def cu_field(field, value, where=None):
        def f(ps, cs, levels):
            i = len(cs) // 2 if where is None else where(cs)
            cs[field][i] = value
        return f

    def first_inter(cs):
        return int(np.flatnonzero(cs["pred_mode"] == 1)[0])

    def ref_idx_out(ps, cs, levels):
        i = first_inter(cs)
        l = 1 if cs["inter_dir"][i] == 1 else 0
        cs["ref_idx"][i, l] = ps["num_ref"][0, l]

    def level_off_out(ps, cs, levels):
        i = int(np.flatnonzero(cs["cbf"][:, 0] != 0)[0])
        cs["level_off"][i, 0] = ps["n_levels"][0] - 3

    def num_ref_out(ps, cs, levels):
        ps["num_ref"][0, 0] = 6

    return [("x outside", cu_field("x", 4096)), ("y negative", cu_field("y", -4)),
            ("x unaligned", cu_field("x", 2)), ("w not a block size", cu_field("w", 12)),
            ("tree", cu_field("tree", 2)), ("inter_dir", cu_field("inter_dir", 3, first_inter)),
            ("ref_idx", ref_idx_out), ("level_off", level_off_out), ("num_ref", num_ref_out)]


@pytest.mark.parametrize("mutation", _mutations(), ids=lambda m: m[0])
def test_plan_rejects_malformed_syntax(mutation):
    """Nothing a stream claims may index outside the cell map, the reference
    table or the level array: Validate refuses the picture (the decode entry
    point then returns XVCGPU_INVALID_ARGUMENT) instead of planning it."""
    fx = sf.StreamFixture("tiny")
    i = next(k for k in range(fx.n) if fx.info[k]["pic_type"] != 2 and fx.info[k]["n_levels"] > 8)
    ps, cs = sf.to_syntax(fx.info[i], fx.cus(i))
    ps, cs = ps.copy(), cs.copy()
    assert decoder.plan_picture(ps, cs, fx.levels(i))[0] >= 1
    mutation[1](ps, cs, fx.levels(i))
    with pytest.raises(ValueError):
        decoder.plan_picture(ps, cs, fx.levels(i))
