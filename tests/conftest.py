import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


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
