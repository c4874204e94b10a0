import importlib.util
import os
import sys

import numpy as np
import pytest

try:                       # torch ships its own HIP / HSA runtime: when a test uses torch on the GPU, that runtime has to be the
    import torch           # one the process loads first (libgsx then binds to it by soname); the other order leaves torch
except ImportError:        # without a device ("No HIP GPUs are available")
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory is `gaussian-splatting_amd` (hyphen), so import it by path."""
    name = "gaussian_splatting_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "gaussian-splatting_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module("gaussian_splatting_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    import orc
    if not orc.Reference.available():
        pytest.skip("oracle/_ref/libgsxref.so not built (reference checkout absent)")
    return orc.Reference()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def gsx(pkg):
    """A live GPU context; tests using it must be marked gpu."""
    g = pkg.Gsx(0)
    yield g
    g.close()


@pytest.fixture
def gsx_fresh(pkg):
    """A context of its own (for tests that change gsx_set_option knobs)."""
    g = pkg.Gsx(0)
    yield g
    g.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
