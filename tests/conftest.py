import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_ck():
    from aquaculture_amd import checkpoint
    return checkpoint.synthetic_checkpoint("yolov5m", 5)


@pytest.fixture(scope="session")
def lib():
    from aquaculture_amd import build, engine
    build.build()
    return engine.load_library()
