import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def vo():
    """The product package; the HIP library must already be built (no fallback)."""
    import visual_odometry_ros_amd as V
    V.load()
    return V


@pytest.fixture(scope="session")
def ctx(vo):
    c = vo.Context(device=0, max_width=1241, max_height=480, max_points=8192, n_slots=4, max_level=6)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx5(vo):
    """BASELINE configs[4]-sized context: 3840x2160 images, 8000+ features."""
    c = vo.Context(device=0, max_width=3840, max_height=2160, max_points=8448, n_slots=3, max_level=4)
    yield c
    c.close()
