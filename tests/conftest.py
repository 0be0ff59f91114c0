import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _native_libraries_up_to_date():
    """Rebuild libpworld.so / the oracle library if a source is newer than the binary (a no-op otherwise).
    The product itself never builds or falls back: without the .so, ``_lib.load()`` raises."""
    import shutil
    from multiagent_rl_amd import build_native
    if shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc'):
        build_native.build()
    from oracle import c_oracle
    c_oracle.build()
    yield
