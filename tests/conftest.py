import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell():
    from raytracer_amd import scenes
    return scenes.cornell_box()


@pytest.fixture(scope="session")
def cornell_env():
    from raytracer_amd import scenes
    return scenes.cornell_box(env=True)


@pytest.fixture(scope="session")
def atrium_small():
    """A ~22 k-triangle version of the Sponza-class atrium (fast on the oracle)."""
    from raytracer_amd import scenes
    return scenes.sponza_class(detail=0.3, env_size=(128, 64))
