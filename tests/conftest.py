import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mlrt():
    """The product package (ctypes binding of libmlrt_hip.so)."""
    # PyTorch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's): in a process that uses both, torch's
    # runtime must be the one that gets loaded, so import torch before libmlrt_hip.so pulls in the system one.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    mod = importlib.import_module("mu-lambda-raytracer_amd")
    if not os.path.exists(mod.LIB_PATH):
        importlib.import_module("mu-lambda-raytracer_amd.build").build()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def earthmap(mlrt):
    return mlrt.load_earthmap()
