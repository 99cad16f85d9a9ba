import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session', autouse=True)
def built_library():
    """libspaa_hip.so is a build product (git-ignored): on a fresh checkout build it once before the first test
    (hipcc cross-compiles gfx950 without a GPU).  Nothing here falls back to a CPU path: if the build is impossible the
    tests that need the library fail loudly."""
    lib = os.path.join(ROOT, 'spaa_amd', 'libspaa_hip.so')
    if not os.path.exists(lib):
        import shutil
        if shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc'):
            import __graft_entry__
            __graft_entry__.build()
    return lib


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions: the largest gate-bookkeeping quantities seen (tests/gates.py: MEASURED), so that its tolerances can be
    stated as a multiple of what is measured."""
    gates = sys.modules.get('gates')
    if gates is not None and any(gates.MEASURED.values()):
        print('\n[gates] largest over this session: ' + ', '.join(f'{k} {v:.3e}' for k, v in gates.MEASURED.items()))
