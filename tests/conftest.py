import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


os.environ.setdefault("EC504_DEBUG_HOOKS", "1")   # arms m1v_debug_fail_alloc (fault injection; inert in a normal process)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref built from /root/reference (authoring container)")


@pytest.fixture(scope="session")
def ref():
    """The real reference (oracle/_ref).  Built on demand where /root/reference exists; tests that
    need it are skipped elsewhere (the GPU box only carries what was prebuilt)."""
    import ref_ffi
    if not ref_ffi.ensure_built():
        pytest.skip("oracle/_ref not available (reference sources absent)")
    return ref_ffi


@pytest.fixture(scope="session")
def orc():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi
