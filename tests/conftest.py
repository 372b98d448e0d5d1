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


def _gpu_available():
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container; GPU tests run on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def rdf():
    return importlib.import_module("3d-beats_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import rdf_oracle
    rdf_oracle.build()
    return rdf_oracle


@pytest.fixture(scope="session")
def oracle_np():
    from oracle import rdf_numpy
    return rdf_numpy


@pytest.fixture()
def host_runtime(rdf, oracle):
    """Installs the host-memory stand-in runtime (tests/fake_runtime.py) for host-logic tests."""
    import fake_runtime
    rt = fake_runtime.HostRuntime()
    prev = rdf.set_runtime(rt)
    yield rt
    rdf.set_runtime(prev)


@pytest.fixture(scope="session")
def gpu_runtime(rdf):
    """The real runtime: torch-ROCm memory + librdf_hip.so.  Fails loudly if either is missing."""
    rdf.set_runtime(None)
    return rdf.get_runtime()
