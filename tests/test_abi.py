"""CPU-side checks of the drop-in boundary: libxvcgpu.so loads without a GPU,
exports every symbol include/xvcgpu.h declares, agrees with the oracle on the
transform tables, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from xvc_amd import api as a
    from xvc_amd import build
    build.build()
    return a


def test_header_symbols_exported(api):
    lib = api.load_library()
    hdr = open(os.path.join(ROOT, "include", "xvcgpu.h")).read()
    declared = set(re.findall(r"\b(xvcgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.xvcgpu_version().decode().endswith("gfx950")


def test_struct_layouts_match_header(api):
    # sizes asserted by both bindings; cross-check field by field
    for a, b in ((api.CU_DTYPE, ol.CU_DTYPE), (api.ME_DTYPE, ol.ME_DTYPE),
                 (api.MERES_DTYPE, ol.MERES_DTYPE), (api.TX_DTYPE, ol.TX_DTYPE),
                 (api.MC_DTYPE, ol.MC_DTYPE)):
        assert a == b
    import oracle_intra
    import oracle_lic
    assert api.INTRA_DTYPE == oracle_intra.INTRA_DTYPE
    assert api.LIC_DTYPE == oracle_lic.LIC_DTYPE
    import oracle_affine_me
    assert api.AFFINE_ME_DTYPE == oracle_affine_me.BLOCK_DTYPE
    assert api.AFFINE_ME_RESULT_DTYPE == oracle_affine_me.RESULT_DTYPE
    src = ("#include <stdio.h>\n#include <stddef.h>\n#include \"xvcgpu.h\"\nint main(){"
           "printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(xvcgpu_cu_info),"
           "sizeof(xvcgpu_me_block), sizeof(xvcgpu_me_result), sizeof(xvcgpu_tx_block),"
           "sizeof(xvcgpu_mc_block), sizeof(xvcgpu_metric_cand),"
           "offsetof(xvcgpu_cu_info, mv), offsetof(xvcgpu_me_block, lambda16),"
           "sizeof(xvcgpu_bi_block), sizeof(xvcgpu_mc_bi_block),"
           "offsetof(xvcgpu_bi_block, boot_mv_x), sizeof(xvcgpu_intra_block),"
           "offsetof(xvcgpu_intra_block, below_left), sizeof(xvcgpu_mc_lic_block),"
           "offsetof(xvcgpu_mc_lic_block, left_x), sizeof(xvcgpu_affine_me_block),"
           "offsetof(xvcgpu_affine_me_block, other_mv), sizeof(xvcgpu_affine_me_result),"
           "sizeof(xvcgpu_copy_segment), sizeof(xvcgpu_frame_pass_args));return 0;}")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"),
                               os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert [int(v) for v in out] == [84, 32, 24, 12, 16, 12, 20, 24, 48, 24, 40, 12, 10, 24, 20,
                                       api.AFFINE_ME_DTYPE.itemsize, 60, api.AFFINE_ME_RESULT_DTYPE.itemsize,
                                       api.SEG_DTYPE.itemsize, C.sizeof(api.FramePassArgs)]
    assert C.sizeof(api.FramePassArgs) == 232
    assert api.AFFINE_ME_DTYPE.itemsize == 84 and api.AFFINE_ME_RESULT_DTYPE.itemsize == 32


def test_new_struct_layouts_match_header(api):
    """xvcgpu_inter_block, the RDOQ records and the parsed-syntax records
    (include/xvc_syntax.h) as the C compiler lays them out."""
    import subprocess
    import tempfile
    from xvc_amd import decoder
    src = ("#include <stdio.h>\n#include <stddef.h>\n#include \"xvcgpu.h\"\n"
           "#include \"xvc_syntax.h\"\nint main(){"
           "printf(\"%zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(xvcgpu_inter_block),"
           "offsetof(xvcgpu_inter_block, mv), sizeof(xvcgpu_rdoq_contexts),"
           "sizeof(xvcgpu_rdoq_params), offsetof(xvcgpu_rdoq_params, ctx_index),"
           "sizeof(xvc_cu_syntax), offsetof(xvc_cu_syntax, mv), sizeof(xvc_picture_syntax));"
           "return 0;}")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"),
                               os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).decode().split()]
    assert out == [api.INTER_DTYPE.itemsize, api.INTER_DTYPE.fields["mv"][1],
                   api.RDOQ_CTX_DTYPE.itemsize, api.RDOQ_PARAMS_DTYPE.itemsize,
                   api.RDOQ_PARAMS_DTYPE.fields["ctx_index"][1],
                   decoder.CU_SYNTAX_DTYPE.itemsize, decoder.CU_SYNTAX_DTYPE.fields["mv"][1],
                   decoder.PICTURE_SYNTAX_DTYPE.itemsize]


def test_transform_tables_match_oracle(api):
    xo = ol.Lib("xo")
    n = 0
    for tx in range(1, 6):
        for size in (2, 4, 8, 16, 32, 64):
            g = api.transform_matrix(tx, size)
            o = xo.transform_matrix(tx, size)
            if o is None:
                assert g is None
                continue
            assert np.array_equal(g, o), (tx, size)
            n += 1
    assert n == 26
    for size in (4, 8, 16, 32):     # XVC_TX_DCT2_LOW: the 6-bit DCT-2 of restricted mode
        assert np.array_equal(api.transform_matrix(7, size), xo.transform_matrix(7, size)), size
    assert api.transform_matrix(7, 8)[1].tolist() == [89, 75, 50, 18, -18, -50, -75, -89]
    assert api.transform_matrix(7, 64) is None and api.transform_matrix(7, 2) is None
    # a few landmark entries of the format (DCT-2 DC row = 256, 4-pt DCT-2)
    assert api.transform_matrix(1, 4).tolist() == [[256, 256, 256, 256],
                                                   [334, 139, -139, -334],
                                                   [256, -256, -256, 256],
                                                   [139, -334, 334, -139]]


def test_no_cpu_fallback_without_device(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.XvcGpuError):
        api.Context(0)


def _build_host_demo(tmpdir):
    import subprocess
    exe = os.path.join(tmpdir, "host_demo")
    subprocess.check_call([
        "g++", "-std=c++11", "-Wall", "-Wextra", "-Werror",
        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "xvc_amd", "host"),
        os.path.join(ROOT, "tests", "host_demo.cc"), "-o", exe,
        "-L", os.path.join(ROOT, "xvc_amd"), "-lxvcgpu",
        "-Wl,-rpath," + os.path.join(ROOT, "xvc_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_layer_compiles_and_refuses_without_gpu(api, tmp_path):
    """The C++ host mirror builds with plain g++ against the C-ABI; without a
    device it reports XVCGPU_NO_DEVICE instead of computing on the CPU."""
    import subprocess
    import torch
    exe = _build_host_demo(str(tmp_path))
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked run")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_layer_runs_on_gpu(api, tmp_path):
    import subprocess
    exe = _build_host_demo(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr



def test_host_headers_compile_together(tmp_path):
    """Every header of the C++ host layer and of include/ in ONE translation unit
    (an integrator following INTEGRATION.md's row-shard snippet includes
    xvc_shard_engine.h and xvc_shard_filter.h together: a name that is a type in
    one and a function in the other would not compile)."""
    import subprocess
    host = os.path.join(ROOT, "xvc_amd", "host")
    inc = os.path.join(ROOT, "include")
    names = sorted(f for f in os.listdir(inc) if f.endswith(".h")) + \
        sorted(f for f in os.listdir(host) if f.endswith(".h"))
    assert "xvc_shard_engine.h" in names and "xvc_shard_filter.h" in names
    src = tmp_path / "all_headers.cc"
    src.write_text("".join('#include "%s"\n' % n for n in names) +
                   "int main() { xvc_shard_plan *p = nullptr; (void)p;\n"
                   "  int (*f)(const int32_t *, int, int, const int32_t *, int32_t *) = "
                   "xvc_shard_filter_plan; (void)f; return 0; }\n")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", inc, "-I", host, str(src)])


def _build_frame_pass_program(tmpdir):
    import subprocess
    host = os.path.join(ROOT, "xvc_amd", "host")
    obj = os.path.join(tmpdir, "synth.o")
    exe = os.path.join(tmpdir, "frame_pass")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-c",
                           os.path.join(host, "xvc_synth.c"), "-o", obj])
    subprocess.check_call([
        "g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror",
        "-I", os.path.join(ROOT, "include"), "-I", host,
        os.path.join(host, "frame_pass_main.cc"), obj, "-o", exe,
        "-L", os.path.join(ROOT, "xvc_amd"), "-lxvcgpu",
        "-Wl,-rpath," + os.path.join(ROOT, "xvc_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_frame_pass_program_builds(api, tmp_path):
    import subprocess
    import torch
    exe = _build_frame_pass_program(str(tmp_path))
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "64", "32", "10", "32", "1"], capture_output=True, text=True)
        assert r.returncode == 3, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_frame_pass_program_matches_oracle(api, tmp_path):
    """The C++ host driver (xvc_amd/host/xvc_frame_pass.h) end to end: three
    chained pictures of the synthetic clip; SSD and a hash of every
    reconstructed plane equal the oracle's."""
    import subprocess
    import numpy as np
    import oracle_frame
    from xvc_amd import pipeline, synth
    exe = _build_frame_pass_program(str(tmp_path))
    pw, ph, bd, qp, frames, BL = 352, 288, 10, 32, 3, 128
    r = subprocess.run([exe, str(pw), str(ph), str(bd), str(qp), str(frames)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == frames

    def fnv(planes):
        h = 14695981039346656037
        for p in planes:
            for b in np.ascontiguousarray(p).astype("<u2").tobytes():
                h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    clip = synth.SyntheticClip(pw, ph, bd)
    pad = lambda planes: [np.ascontiguousarray(np.pad(p, BL if c == 0 else BL // 2,
                                                      mode="edge")) for c, p in enumerate(planes)]
    desc = pipeline.FrameDescriptors(pw, ph, qp)
    ref = pad(clip.frame(0))
    xo = ol.Lib("xo")
    for n in range(1, frames + 1):
        rec, _, _, _, ssd = oracle_frame.frame_pass(desc, bd, pad(clip.frame(n)), ref, BL,
                                                    ref_poc=n - 1, lib=xo)
        inner = [rec[c][(BL >> (c > 0)):(BL >> (c > 0)) + (ph >> (c > 0)),
                        (BL >> (c > 0)):(BL >> (c > 0)) + (pw >> (c > 0))] for c in range(3)]
        got = lines[n - 1]
        assert (int(got[3]), int(got[5])) == ssd, n
        assert int(got[7], 16) == fnv(inner), n
        ref = rec


def test_cpp_builder_composes_a_state(api, tmp_path):
    """A caller written in C++ hands ONE CompressInter state to xvc_gpu::CuStateBuilder
    (xvc_cu_state_builder.h: what CuEncoder holds - CU, reference lists, AMVP predictors per
    (list, picture), lambda, contexts) and gets the pass record, the work arrays and the
    chain's op program: built with plain g++ against libxvchost.so, no GPU needed."""
    import subprocess
    from xvc_amd import build
    build.build_host()
    src = tmp_path / "one_state.cc"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include "xvc_cu_state_builder.h"
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
int main() {
  // a 16x16 CU at (32, 16) of a B picture with one picture per list (POC 0 and POC 4)
  xvc_cs_state st;
  std::memset(&st, 0, sizeof(st));
  st.kind = XVC_CS_INTER; st.x = 32; st.y = 16; st.w = 16; st.h = 16; st.supported = 1;
  st.me_first = 0; st.me_count = 2; st.ev = 0; st.call_first = 0; st.call_pass0 = 3;
  st.merge = -1; st.in_satd = -1; st.level_count = 384;
  xvc_csb_ref_entry en[2];
  std::memset(en, 0, sizeof(en));
  en[0].list = 0; en[0].ref_idx = 0; en[0].mvp[0][0][0] = 16; en[0].mvp[1][0][1] = -32;
  en[1].list = 1; en[1].ref_idx = 0; en[1].mvp[0][0][0] = -16;
  xvc_csb_motion mo;
  std::memset(&mo, 0, sizeof(mo));
  mo.state = 0; mo.nb = -1; mo.plain.first = 0; mo.plain.n = 2; mo.plain.lambda16 = 123456;
  xvcgpu_me_block me[2];
  std::memset(me, 0, sizeof(me));
  for (int k = 0; k < 2; k++) { me[k].x = 32; me[k].y = 16; me[k].w = me[k].h = 16; me[k].search_range = 64; }
  const int8_t me_ref[2] = {0, 1};
  const int32_t slot_pocs[2] = {0, 4};
  xvcgpu_inter_block ev_inter[3];
  std::memset(ev_inter, 0, sizeof(ev_inter));
  xvc_csb_eval ev;
  std::memset(&ev, 0, sizeof(ev));
  ev.x = 32; ev.y = 16; ev.merge_slot = -1;
  for (int c = 0; c < 3; c++) ev.weight[c] = 1.0;
  const int32_t ev_ctx[1] = {0};
  xvcgpu_metric_cand call_cand[3];
  std::memset(call_cand, 0, sizeof(call_cand));
  const uint8_t call_comp[3] = {0, 1, 2};
  const int32_t call_ev[3] = {0, 0, 0};
  xvc_csb_picture pic;
  std::memset(&pic, 0, sizeof(pic));
  pic.states = &st; pic.n_states = 1;
  pic.ref_poc[0][0] = 0; pic.ref_poc[1][0] = 4; pic.n_ref[0] = pic.n_ref[1] = 1;
  pic.slot_pocs = slot_pocs; pic.n_slots = 2; pic.lic_folds = 1;
  pic.motions = &mo; pic.n_motions = 1; pic.entries = en;
  pic.me_jobs = me; pic.me_ref = me_ref; pic.n_me = 2;
  pic.ev_inter = ev_inter; pic.n_ev = 1; pic.evals = &ev; pic.ev_ctx = ev_ctx;
  pic.call_cand = call_cand; pic.call_comp = call_comp; pic.call_ev = call_ev; pic.n_calls = 3;
  xvc_csb *b = nullptr;
  CHECK(xvc_host_csb_build(&pic, &b) == 0 && b);
  int64_t bytes = 0;
  const xvcgpu_cs_pass *ps = (const xvcgpu_cs_pass *)xvc_host_csb_array(b, XVC_CSB_PASSES, &bytes);
  CHECK(bytes == (int64_t)sizeof(xvcgpu_cs_pass) && ps);
  CHECK(ps->x == 32 && ps->w == 16 && ps->lambda16 == 123456 && ps->eval == 0 && ps->plain_pass == -1);
  CHECK(ps->uni_job[0][0] == 0 && ps->uni_job[1][0] == 1 && ps->slot[0][0] == 0 && ps->slot[1][0] == 1);
  CHECK(ps->mvp[0][0][1][0][1] == -32 && ps->same_poc_in_l0[0] == -1 && ps->bi_job == 0);
  CHECK(xvc_host_csb_n_start_dist(b) == 4 && xvc_host_csb_n_bi_slots(b) == 2 * 3 * 3);
  const xvcgpu_me_block *mw = (const xvcgpu_me_block *)xvc_host_csb_array(b, XVC_CSB_ME_WORK, &bytes);
  CHECK(bytes == 2 * (int64_t)sizeof(xvcgpu_me_block) && mw[0].mvp_x == 0x7fffff);   // the device's to compose
  xvc_csb_addrs a;
  std::memset(&a, 0, sizeof(a));
  xvc_csb_intra in;
  std::memset(&in, 0, sizeof(in));
  int64_t n_ops = 0;
  const xvc_cs_op *ops = xvc_host_csb_program(
      b, &a, &in, 0, 1, XVC_CSB_BY_POSITION | XVC_CSB_REFS_FORM | XVC_CSB_NO_COPIES | XVC_CSB_FUSED_EVAL, &n_ops);
  const int want[] = {XVC_OP_MC_METRIC_REFS, XVC_OP_START_FOLD, XVC_OP_ME_REFS, XVC_OP_UNI_FOLD,
                      XVC_OP_BI_REFS, XVC_OP_BI_FOLD, XVC_OP_INTER_PRED, XVC_OP_RESIDUAL};
  CHECK(ops && n_ops >= 9);
  for (int k = 0; k < 8; k++) CHECK(ops[k].opcode == want[k]);
  CHECK(ops[0].n == 4 && ops[2].n == 2 && ops[2].i0 == 16 && ops[7].n == 3 && ops[7].r0 == 3);
  CHECK(ops[n_ops - 1].opcode == XVC_OP_SYNC && ops[n_ops - 1].i0 == 1);
  // an entry list that does not cover both lists is refused
  mo.plain.n = 1;
  xvc_csb *b2 = nullptr;
  CHECK(xvc_host_csb_build(&pic, &b2) != 0 && !b2);
  xvc_host_csb_destroy(b);
  std::printf("OK %lld ops\n", (long long)n_ops);
  return 0;
}
''')
    exe = str(tmp_path / "one_state")
    host = os.path.join(ROOT, "xvc_amd")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(host, "host"), str(src), "-o", exe, "-L", host, "-lxvchost",
                           "-lxvcgpu", "-Wl,-rpath," + host])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_cu_state_layouts_match_the_host_header():
    """xvc_amd/cu_state.py (the walk's Python host side) mirrors the structs of
    xvc_amd/host/xvc_cu_state.h: sizes and a late field of each, against g++."""
    import subprocess
    import tempfile
    from xvc_amd import cu_state as cs
    src = ("#include <stdio.h>\n#include <stddef.h>\n#include \"xvc_cu_state.h\"\nint main(){"
           "printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(xvc_cs_state),"
           "offsetof(xvc_cs_state, level_first), sizeof(xvc_cs_tables), offsetof(xvc_cs_tables, h_in_levels),"
           "sizeof(xvc_cs_stats), sizeof(xvc_cs_op), sizeof(xvc_cs_env), offsetof(xvc_cs_env, d_in_levels),"
           "sizeof(xvcgpu_cs_result));return 0;}")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cc"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "xvc_amd", "host"),
                               os.path.join(d, "t.cc"), "-o", os.path.join(d, "t")])
        out = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).decode().split()]
    assert out == [cs.STATE_DTYPE.itemsize, cs.STATE_DTYPE.fields["level_first"][1],
                   C.sizeof(cs.CsTables), cs.CsTables.h_in_levels.offset, C.sizeof(cs.CsStats),
                   cs.OP_DTYPE.itemsize, C.sizeof(cs.CsEnv), cs.CsEnv.d_in_levels.offset,
                   cs.RESULT_DTYPE.itemsize]
