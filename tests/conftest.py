import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native libraries: prebuilt in-tree (they travel with the snapshot); build if missing."""
    from ray_tracing_in_one_weekend_amd import build as b
    b.build_host_library()
    from oracle import build as ob
    ob.build_oracle()
    if not os.path.exists(os.path.join(b.PKG_DIR, "librtow_mi355x.so")):
        b.build_gpu_library()
    return True


@pytest.fixture(scope="session")
def rt(built):
    import ray_tracing_in_one_weekend_amd as rt
    rt.register_default_images()
    return rt


@pytest.fixture(scope="session")
def orc(built):
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def renderer(rt):
    r = rt.Renderer(0)
    yield r
    r.close()


@pytest.fixture(autouse=True)
def _default_options(request):
    """The session's Renderer goes back to the library's own choices after every test (rt_debug_set_option is per context,
    and a test that fails between setting and resetting an option must not colour the ones behind it)."""
    yield
    if "renderer" in request.fixturenames:
        r = request.getfixturevalue("renderer")
        from ray_tracing_in_one_weekend_amd import _ffi
        for opt in _ffi.OPT_NAMES.values():
            r.set_option(opt, 0)
