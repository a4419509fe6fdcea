import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "stages.npz"))


@pytest.fixture(scope="session")
def synth_sd(manifest):
    """Full synthetic state dict: reference-manifest tensors + the CompressAI-side ones."""
    from dc_vic_amd.synth import full_synth_state_dict
    return full_synth_state_dict(seed=1234)
