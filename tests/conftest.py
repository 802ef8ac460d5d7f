import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    from raytracer_amd import rtg
    ctx = rtg.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(autouse=True)
def _restore_ctx_options(request):
    """Context options are state of the shared session context: whatever a test sets (or leaves set when it fails
    half way) is put back to the library defaults before the next test runs."""
    yield
    if "gpu_ctx" in request.fixturenames:
        ctx = request.getfixturevalue("gpu_ctx")
        for key, value in (("traversal", 2), ("pretest", 2), ("frame_bands", 2), ("time_kernels", 1), ("stats", 1), ("sort_rays", 1), ("keep_state", 1), ("comm_loopback", 1), ("photon_trig", 1), ("kd_finish", 2), ("photon_batch", 23),
                           ("tri_min", 20), ("obj_min", 4), ("patch_tri_min", 24), ("max_blocks", 0)):
            ctx.set_option(key, value)
        ctx.set_stream(None)
