import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a GPU test that takes minutes (still part of -m gpu; deselect with -m 'gpu and not slow')")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor): the parity metric of SURVEY 8(c) (floor |u| >= 1e-3)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.fixture
def tuned():
    """tuned(**knobs): override scheduling knobs of the library (aoc_set_tuning) for the rest of the test; every
    call starts again from the settings the test began with, which are restored at the end."""
    import ctypes as C
    from aircraftoptimalcontrol_amd import _lib
    l = _lib.lib()
    old = _lib.Tuning()
    l.aoc_get_tuning(C.byref(old))

    def set_(**kw):
        new = _lib.Tuning.from_buffer_copy(old)
        for k, v in kw.items():
            assert k in dict(_lib.Tuning._fields_) and not k.startswith("reserved"), k
            setattr(new, k, int(v))
        l.aoc_set_tuning(C.byref(new))
        return new

    yield set_
    l.aoc_set_tuning(C.byref(old))
