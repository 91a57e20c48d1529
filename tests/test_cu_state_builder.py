"""xvc_gpu::CuStateBuilder (xvc_amd/host/xvc_cu_state_builder.{h,cc}): the C++ composer of
the CU-state walk against digests of what the Python composer of rounds 4 - 5 produced from
the same captured pictures (tests/golden/cs_builder_digests.json, written by
tools/gen_cs_builder_digests.py while both composers existed and agreed byte for byte), plus
structural checks that do not need the digests.  CPU only: the builder is host code."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import rd_serial
from xvc_amd import api, cu_state_builder as csb

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cs_builder_digests.json")
# (by_position, verify, refs_form, live, no_copies, fused_eval, merge_fold)
VARIANTS = [(1, 1, 1, 0, 1, 1, 1), (0, 1, 1, 0, 1, 1, 1), (1, 0, 1, 1, 1, 1, 1), (1, 1, 0, 0, 0, 0, 0),
            (1, 1, 1, 0, 1, 0, 1), (1, 1, 0, 1, 0, 0, 1), (1, 0, 1, 0, 0, 0, 0), (0, 0, 1, 1, 1, 0, 0)]


def fake_addrs(sp):
    """Distinct, fixed base addresses for every array a program refers to (a program is a
    function of the addresses; the digests were taken with these)."""
    a = csb.Addrs()
    for k, f in enumerate(csb.ADDR_FIELDS):
        setattr(a, f, 0x10000000 * (k + 2))
    keep = dict(in_stage=np.ascontiguousarray(sp.in_stage, np.int32),
                in_ctx=np.ascontiguousarray(sp.in_ctx, np.int32),
                in_comp=np.ascontiguousarray(sp.in_comp, np.int32),
                in_weight=np.ascontiguousarray(sp.in_weight, np.float64),
                in_off=np.ascontiguousarray(sp.in_off, np.uint32),
                bi_ref=np.ascontiguousarray(sp.bi_ref, np.int8))
    i = csb.Intra()
    for name, v in keep.items():
        setattr(i, name, v.ctypes.data if v.size else None)
    i.n_in, i.n_in_levels = len(sp.in_off), int(sp.n_in_levels)
    return a, i, keep


def flags_of(v):
    byp, ver, refs, live, nc, fe, mf = v
    return (byp * csb.BY_POSITION | ver * csb.VERIFY | refs * csb.REFS_FORM | live * csb.LIVE |
            nc * csb.NO_COPIES | fe * csb.FUSED_EVAL | mf * csb.MERGE_FOLD)


def digests(name, poc):
    sp = rd_serial.SerialPicture(api, name, poc)
    b = rd_serial.compose(sp, rd_serial.ref_lists_of(name, poc))
    out = {"n_start_dist": b.n_start_dist, "n_bi_slots": b.n_bi_slots, "n_edist": int(b.n_edist)}
    for arr, _ in csb.ARRAYS:
        out[arr] = hashlib.sha256(getattr(b, arr).tobytes()).hexdigest()
    a, i, keep = fake_addrs(sp)
    n_st = len(sp.states)
    for v in VARIANTS:
        for first, n in ((0, n_st), (n_st // 3, min(2000, n_st - n_st // 3))):
            ops = b.program(a, i, first, n, flags_of(v))
            out["program_%s_%d_%d" % ("".join(map(str, v)), first, n)] = \
                [len(ops), hashlib.sha256(ops.tobytes()).hexdigest()]
    return out, sp, b


@pytest.mark.parametrize("name,poc", [("tiny", 2), ("c0", 2), ("c0", 4)])
def test_builder_output_equals_the_recorded_composition(name, poc):
    want = json.load(open(GOLDEN))["%s_%d" % (name, poc)]
    got, sp, b = digests(name, poc)
    bad = [k for k in want if got.get(k) != want[k]]
    assert not bad, bad[:8]
    b.destroy()


def test_program_structure():
    """What any program must satisfy, whatever the digests say: every chain ends in a SYNC
    whose count is the number of states it holds, read-backs come in front of their SYNC, a
    fold's pass index lies in its state's range, by-state programs hold one state per chain,
    live chains never hold two inter states."""
    got, sp, b = digests("tiny", 2)
    a, i, keep = fake_addrs(sp)
    st = sp.states
    n_sup = int((st["supported"] != 0).sum())
    for v in VARIANTS:
        ops = b.program(a, i, 0, len(st), flags_of(v))
        assert ops["opcode"][-1] == rd_serial.OP_SYNC
        syncs = ops[ops["opcode"] == rd_serial.OP_SYNC]
        assert int(syncs["i0"].sum()) == n_sup          # (inner waits carry 0)
        if not v[0] and not v[3]:                       # a chain per state
            assert (syncs["i0"] <= 1).all()
        folds = ops[np.isin(ops["opcode"], (rd_serial.OP_START_FOLD, rd_serial.OP_UNI_FOLD,
                                            rd_serial.OP_BI_FOLD))]
        assert (folds["i0"] >= 0).all() and (folds["i0"] < len(b.passes)).all()
        # three folds per pass, in order
        assert len(folds) == 3 * len(b.passes)
        assert (folds["i0"].reshape(-1, 3) == folds["i0"][::3, None]).all()
        fetch = ops["opcode"] == rd_serial.OP_FETCH
        assert (ops["n"][fetch] > 0).all()
    b.destroy()


def test_builder_refuses_what_the_folds_do_not_run():
    """A pass whose entries do not cover both lists is refused (the folds index
    [list][ref_idx] over the picture's lists)."""
    sp = rd_serial.SerialPicture(api, "tiny", 2)
    sp.ev_merge_slot = rd_serial.match_merge_slots(sp)
    inp = rd_serial.builder_inputs(sp, rd_serial.ref_lists_of("tiny", 2))
    m = inp["motions"].copy()
    k = int(np.flatnonzero(m["plain"]["n"] > 0)[0])
    m["plain"]["n"][k] -= 1
    inp["motions"] = m
    with pytest.raises(ValueError):
        csb.Builder(inp)


def test_struct_sizes_match_the_header():
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.cc")
        open(src, "w").write(
            '#include <cstdio>\n#include "xvc_cu_state_builder.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu '
            '%zu %zu %zu\\n", sizeof(xvc_csb_ref_entry), sizeof(xvc_csb_pass_in), sizeof(xvc_csb_motion), '
            'sizeof(xvc_csb_neighbours), sizeof(xvc_csb_merge), sizeof(xvc_csb_eval), sizeof(xvc_csb_picture), '
            'sizeof(xvc_csb_addrs), sizeof(xvc_csb_intra));return 0;}\n')
        exe = os.path.join(td, "sz")
        subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(root, "include"), "-I",
                               os.path.join(root, "xvc_amd", "host"), src, "-o", exe])
        got = tuple(map(int, subprocess.check_output([exe]).split()))
    want = (csb.REF_ENTRY_DTYPE.itemsize, csb.PASS_IN_DTYPE.itemsize, csb.MOTION_DTYPE.itemsize,
            csb.NEIGHBOURS_DTYPE.itemsize, csb.MERGE_DTYPE.itemsize, csb.EVAL_DTYPE.itemsize,
            C.sizeof(csb.Picture), C.sizeof(csb.Addrs), C.sizeof(csb.Intra))
    assert got == want, (got, want)
