import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/cv_oracle.py); builds liborc.so on first use."""
    from oracle import cv_oracle
    cv_oracle.lib()
    return cv_oracle


@pytest.fixture(scope="session")
def golden_calls():
    import json
    z = np.load(os.path.join(GOLDEN, "getmaps_calls.npz"))
    with open(os.path.join(GOLDEN, "getmaps_calls.json")) as f:
        meta = json.load(f)
    return z, meta


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container (GPU tests run on the MI355X box)")
    import gyroflow_amd
    gyroflow_amd._lib.load()
    return torch.device("cuda", 0)


def preset_path(key):
    return os.path.join(GOLDEN, "presets", key + ".json")
