"""InterSearch::GetInterPredBits with the encoder's DEFAULT setting
(fast_inter_pred_bits == 0: CuWriter::WriteInterPrediction through a throw-away
RdoSyntaxWriter on the live CABAC state, inter_search.cc:1131-1135) as the product
computes it from a snapshot of eleven context states (include/xvc_inter_bits.h):
every candidate SearchRefIdx priced while the reference encoder coded the tiny clip
(all pictures), the CIF clip = BASELINE config 0 (all pictures) and one 1080p B picture - uni-directional, list-1 re-use,
bi-directional, affine (tests/golden/rd_order_*.npz) - gets the reference's bits."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import order_fixture as of
from xvc_amd import decoder

SYNTAX_DTYPE = np.dtype([
    ("inter_dir", "u1"), ("use_affine", "u1"), ("fullpel_mv", "u1"), ("use_lic", "u1"),
    ("ref_idx", "i1", 2), ("mvp_idx", "u1", 2), ("force_mvd_zero", "u1", 2),
    ("reserved", "u1", 2), ("mvd", "<i4", (2, 2, 2))])


def mvd_of(mv, mvp, fullpel):
    """MotionVector - MotionVector -> MvDelta (cu_types.h:192-194: quarter samples),
    then InterSearch::SetMvd's extra shift for whole-sample vectors (:1026-1031)."""
    d = (mv - mvp) >> 2
    return np.where(fullpel[:, None], d >> 2, d)


def syntax_of(cd):
    """The CU's inter state while SearchRefIdx prices candidate cd[i]."""
    n = len(cd)
    s = np.zeros(n, SYNTAX_DTYPE)
    i = np.arange(n)
    lst = cd["list"].astype(np.int64)
    oth = 1 - lst
    affine = cd["kind"] >= 2
    fullpel = (cd["flags"] & 1) != 0
    s["inter_dir"] = cd["inter_dir"]
    s["use_affine"] = affine
    s["fullpel_mv"] = fullpel
    s["use_lic"] = (cd["flags"] & 2) != 0
    s["ref_idx"][i, lst] = cd["ref_idx"]
    s["mvp_idx"][i, lst] = cd["mvp_idx"]
    mvp = cd["mvp"][i, cd["mvp_idx"].astype(np.int64)]           # [n, 3, 2]
    for k in range(2):
        s["mvd"][i, lst, k] = mvd_of(cd["mv"][:, k], mvp[:, k], fullpel)
    s["mvd"][~affine, :, 1] = 0
    bi = cd["inter_dir"] == 2
    s["ref_idx"][i[bi], oth[bi]] = cd["other_ref_idx"][bi]
    s["mvp_idx"][i[bi], oth[bi]] = cd["other_mvp_idx"][bi]
    s["force_mvd_zero"][i[bi], oth[bi]] = cd["force_mvd_zero_other"][bi]
    s["mvd"][i[bi], oth[bi]] = cd["other_mvd"][bi]
    return s


@pytest.fixture(scope="module")
def host():
    L = decoder.load_host_library()
    L.xvc_host_inter_pred_bits.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    L.xvc_host_inter_pred_bits.restype = None
    return L


@pytest.mark.parametrize("name", ["tiny", "c0", "c1"])
def test_bits_of_every_priced_candidate(host, name):
    o = of.load(name)
    cd, ictx = o["cands"], np.ascontiguousarray(o["ictx"])
    assert len(cd) > 50000 and ictx.dtype.itemsize == 16
    syn = np.ascontiguousarray(syntax_of(cd))
    idx = np.ascontiguousarray(cd["ictx_index"], np.int32)
    bits = np.zeros(len(cd), np.uint32)
    host.xvc_host_inter_pred_bits(ictx.ctypes.data, idx.ctypes.data, syn.ctypes.data, len(cd),
                                  bits.ctypes.data)
    bad = np.flatnonzero(bits != cd["bits"])
    assert not len(bad), (len(bad), cd[bad[:3]], bits[bad[:3]])
    # the closed form of restricted mode (:1084-1130) is NOT what the default run pays
    kinds = {int(k): int((cd["kind"] == k).sum()) for k in np.unique(cd["kind"])}
    assert kinds.get(0, 0) and kinds.get(1, 0)
    if name != "tiny":
        assert kinds.get(2, 0) and kinds.get(3, 0)          # affine, uni and bi
    assert (cd["reused"] != 0).any() and ((cd["flags"] & 1) != 0).any()


def test_context_state_machine_equals_reference(host):
    """The MPS / LPS transitions the product derives from the CABAC rule against the
    reference's two 128-entry tables (context_model.cc:51-73), and the bin costs."""
    if not ol.have_ref():
        pytest.skip("reference build not present")
    ref = C.CDLL(ol.REF_SO)
    ref.xr_next_state_table.restype = C.POINTER(C.c_uint8)
    ref.xr_entropy_bits_table.restype = C.POINTER(C.c_uint32)
    host.xvc_host_entropy_bits_table.restype = C.POINTER(C.c_uint32)
    for lps in (0, 1):
        got = np.zeros(128, np.uint8)
        host.xvc_host_next_state_table(lps, got.ctypes.data_as(C.c_void_p))
        want = np.ctypeslib.as_array(ref.xr_next_state_table(lps), (128,))
        assert np.array_equal(got, want), lps
    assert np.array_equal(np.ctypeslib.as_array(host.xvc_host_entropy_bits_table(), (128,)),
                          np.ctypeslib.as_array(ref.xr_entropy_bits_table(), (128,)))
