import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'optimized-diffusion-model_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def params0():
    from oracle.weights import make_params
    return make_params(0)


@pytest.fixture(scope='module')
def emu():
    """Bind rdmi to the CPU emulator build of the SAME csrc sources for this test module, then restore.
    (Test infrastructure: the product only ever loads librdmi.so.)"""
    from tests.emu.build_emu import build
    from rdmi import _native
    prev = (_native._lib, _native._lib_path)
    # RDMI_EMU_ASAN=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0): the AddressSanitizer build
    _native.use_library(build(sanitize=bool(os.environ.get('RDMI_EMU_ASAN'))))
    assert _native.is_emulator()
    yield _native
    _native._lib, _native._lib_path = prev
