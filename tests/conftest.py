import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def _have_gpu():
    """True when this box has an AMD GPU the tests can run on.

    Only the ABSENCE of a device skips the GPU tests: no /dev/kfd (the authoring
    container).  Where a device node exists - or XVC_REQUIRE_GPU=1 says the run
    must be a GPU run - every failure to reach it (library not built, dlopen or
    ABI error, xvcgpu_create refusing the device) is raised, so that a broken
    libxvcgpu.so cannot turn into a green run with zero GPU coverage."""
    required = os.environ.get("XVC_REQUIRE_GPU", "0") == "1"
    if not os.path.exists("/dev/kfd") and not required:
        return False
    from xvc_amd import api
    ctx = api.Context(0)
    ctx.close()
    return True


def pytest_collection_modifyitems(config, items):
    """`pytest tests/` on a box without a gfx950 device skips the GPU tests
    instead of erroring in their fixtures (-m gpu on a GPU box runs them all)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _have_gpu():
        return
    skip = pytest.mark.skip(reason="no gfx950 device (xvcgpu_create failed)")
    for it in gpu_items:
        it.add_marker(skip)


# Soak runs (tools/soak.sh): XVC_SOAK=<k> shifts every seeded generator of the
# randomised parity tests, so that the same tests see fresh data.  The golden
# vector tests do not draw random numbers and are unaffected.
_SOAK = int(os.environ.get("XVC_SOAK", "0"))
if _SOAK:
    import numpy as _np
    _orig_rng = _np.random.default_rng

    def _shifted(seed=None, *a, **k):
        if isinstance(seed, (int, _np.integer)):
            seed = int(seed) + 7919 * _SOAK
        return _orig_rng(seed, *a, **k)

    _np.random.default_rng = _shifted
