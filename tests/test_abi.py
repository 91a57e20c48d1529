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
    src = ("#include <stdio.h>\n#include <stddef.h>\n#include \"xvcgpu.h\"\nint main(){"
           "printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(xvcgpu_cu_info),"
           "sizeof(xvcgpu_me_block), sizeof(xvcgpu_me_result), sizeof(xvcgpu_tx_block),"
           "sizeof(xvcgpu_mc_block), sizeof(xvcgpu_metric_cand),"
           "offsetof(xvcgpu_cu_info, mv), offsetof(xvcgpu_me_block, lambda16),"
           "sizeof(xvcgpu_bi_block), sizeof(xvcgpu_mc_bi_block),"
           "offsetof(xvcgpu_bi_block, boot_mv_x));return 0;}")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"),
                               os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert [int(v) for v in out] == [84, 32, 24, 12, 16, 12, 20, 24, 48, 24, 40]


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
