"""The CU-state tables of the serial RD replay without a device: the bindings' struct
sizes, and the grouping of a captured encode into states (tests/rd_serial.py)."""
import ctypes as C

import numpy as np

import rd_serial
from xvc_amd import api, decoder


def test_binding_sizes():
    L = decoder.load_host_library()
    out = (C.c_int32 * 7)()
    L.xvc_host_cs_sizes(out)
    assert list(out) == [rd_serial.STATE_DTYPE.itemsize, C.sizeof(rd_serial.CsTables),
                         C.sizeof(rd_serial.CsStats), rd_serial.PASS_DTYPE.itemsize,
                         rd_serial.RESULT_DTYPE.itemsize, rd_serial.OP_DTYPE.itemsize,
                         C.sizeof(rd_serial.CsEnv)]


def test_host_library_exports_the_cu_state_entry_points():
    L = decoder.load_host_library()
    for name in ("xvc_host_cu_state_run_serial", "xvc_host_cs_run_program",
                 "xvc_host_cs_run_programs_interleaved", "xvc_host_inter_pred_bits",
                 "xvc_host_next_state_table"):
        assert getattr(L, name) is not None
    # argument checks need no device
    L.xvc_host_cs_run_programs_interleaved.restype = C.c_int
    assert L.xvc_host_cs_run_programs_interleaved(0, None, None, None, None, None) != 0


def test_states_of_a_captured_picture():
    sp = rd_serial.SerialPicture(api, "tiny", 2)
    st, s = sp.states, sp.summary()
    assert s["states"] > 5000 and s["inter"] > 1000 and s["merge_rank"] > 500
    # every record of the picture sits in exactly one state, in capture order
    assert s["me"] == int((sp.me["poc"] == 2).sum())
    steps = sp.rd["steps"]
    assert s["bi"] + s["affine"] == int((steps["poc"] == 2).sum())
    ev = sp.rd["evals"]
    calls = sp.rd["calls"]
    assert s["calls"] == int((ev["poc"][calls["eval"]] == 2).sum())
    for f, c in (("me_first", st["me_count"]), ("bi_first", st["bi_count"]),
                 ("aff_first", st["aff_uni_count"] + st["aff_bi_count"]),
                 ("call_first", st["call_pass0"] + st["call_pass1"])):
        used = c > 0
        assert np.array_equal(st[f][used], np.r_[0, np.cumsum(c[used])[:-1]])
    # an inter state: searches, then an evaluation whose motion is one the search priced
    inter = st[st["kind"] == rd_serial.KIND_INTER]
    assert (inter["me_count"] > 0).all() and (inter["ev"] >= 0).all() and (inter["cand_count"] > 0).all()
    # the state's vector is its SearchMotion's final choice (first pass, or the affine second)
    fin = sp.order["finals"]
    n_checked = 0
    for r in inter[:400]:
        e = sp.ev_want[r["ev"]]
        fs = fin[int(r["final_first"]):int(r["final_first"]) + int(r["final_count"])]
        assert len(fs) in (1, 2)
        hit = [f for f in fs if f["inter_dir"] == e["inter_dir"] and
               all(np.array_equal(f["mv"][l], e["mv"][l]) for l in range(2)
                   if e["inter_dir"] == 2 or e["inter_dir"] == l)]
        assert hit, (r, e, fs)
        n_checked += 1
    assert n_checked == 400


def test_passes_of_a_captured_picture():
    """Every SearchMotion of the picture becomes a pass whose entries are the (list,
    picture) pairs SearchRefIdx walked; the affine pass follows its plain pass."""
    sp = rd_serial.SerialPicture(api, "tiny", 2)
    rd_serial.compose(sp, rd_serial.ref_lists_of("tiny", 2))
    st, ps = sp.states, sp.passes
    # (the LIC states too: XVC_CS_LIC passes, their neighbour records are in the capture)
    motion = ((st["kind"] == rd_serial.KIND_INTER) | (st["kind"] == rd_serial.KIND_MOTION)) & (st["supported"] != 0)
    assert np.array_equal(motion, sp.folded)
    assert ((ps["flags"] & rd_serial.CS_LIC) != 0).sum() == ((st["flags"] & rd_serial.STATE_LIC) != 0)[motion].sum() > 2000
    assert (sp.pass_count[motion] >= 1).all() and (sp.pass_count[~motion] == 0).all()
    assert len(ps) == int(sp.pass_count.sum()) > 1000
    aff = (ps["flags"] & rd_serial.CS_AFFINE) != 0
    assert aff.any() and (ps["plain_pass"][aff] == np.flatnonzero(aff) - 1).all()
    assert (ps["plain_pass"][~aff] == -1).all()
    # the searches of a pass are the state's jobs, each once
    used = ps["uni_job"][~aff]
    used = np.sort(used[used >= 0])
    sup = st[motion]
    want = np.concatenate([np.arange(a, a + c) for a, c in zip(sup["me_first"], sup["me_count"])])
    assert np.array_equal(used, want)


def test_merge_fold_records_of_a_captured_picture():
    """xvc_gpu::CuStateBuilder's merge folds: one xvcgpu_cs_merge record and four evaluation
    slots per merge ranking; the captured merge-candidate evaluations find their slot
    (ranked candidate with the same motion) - the harness-side map the GPU test checks the
    device's fold through.  Struct sizes as the C compiler lays them out."""
    import subprocess, os, tempfile
    # (tiny's rankings mostly hold an illumination-compensated candidate: those rankings
    # are not replayed, their plain candidates' evaluations keep the capture's job)
    for name, poc, least in (("tiny", 2, 0.5), ("c0", 4, 0.85)):
        sp = rd_serial.SerialPicture(api, name, poc)
        rd_serial.compose(sp, rd_serial.ref_lists_of(name, poc))
        n_m = len(sp.mg_inter)
        assert len(sp.mg_fold) == n_m and sp.mg_slots.shape == (4 * n_m, 3)
        st = sp.states
        ev = st["ev"][(st["kind"] == rd_serial.KIND_EVAL) & (st["supported"] != 0)]
        mapped = sp.ev_merge_slot[ev] >= 0
        assert mapped.mean() > least, (name, mapped.mean())
        # a slot is used by at most one evaluation, and lies inside its ranking's count
        used = sp.ev_merge_slot[sp.ev_merge_slot >= 0]
        assert len(np.unique(used)) == len(used)
        assert ((used % 4) < sp.mg_want["num"][used // 4]).all()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "xvcgpu_types.h"\n'
                             'int main(){printf("%zu %zu\\n", sizeof(xvcgpu_cs_merge), '
                             'sizeof(xvcgpu_cs_merge_result));return 0;}\n')
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), src, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert (a, b) == (rd_serial.MERGE_FOLD_DTYPE.itemsize, rd_serial.MERGE_RESULT_DTYPE.itemsize)


def test_walked_stretch_has_the_pictures_mix_of_states():
    """tools/cu_state_walk.py and bench.py walk the stretch SerialPicture.representative_start
    picks: the 1080p picture's first 4000 states hold 184 of its 220 intra states (58
    TransformAndReconstruct calls each), the picked stretch the picture's share of them."""
    sp = rd_serial.SerialPicture(api, "c1", 2)
    st = sp.states
    n = 4000
    intra = st["kind"] == rd_serial.KIND_INTRA
    assert intra.sum() == 220 and intra[:n].sum() > 150
    a = sp.representative_start(n)
    want = intra.sum() * n / len(st)
    assert 0.5 * want <= intra[a:a + n].sum() <= 2 * want, (a, intra[a:a + n].sum(), want)
    # a stretch starts where a visit of a CU position starts
    assert a == sp.position_start(a)
    for k in range(4):
        share, whole = (st["kind"][a:a + n] == k).mean(), (st["kind"] == k).mean()
        assert abs(share - whole) < 0.03, (k, share, whole)
