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
    # the oracle's fp32 sums depend on torch's CPU thread count: pinned here so that one test file run alone sees the same oracle as
    # the whole suite (where tests/test_oracle_golden.py used to set it as an import side effect)
    import torch as _torch
    _torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    if os.environ.get("DCVIC_POISON_EMPTY"):
        # diagnostic run: every float tensor torch.empty / empty_like hands out on the GPU is filled with NaN, so a kernel that reads
        # an element nobody wrote (stale data of the caching allocator -- results that depend on what ran before) fails loudly
        import torch
        real_empty, real_empty_like = torch.empty, torch.empty_like

        def poisoned(t):
            if t.is_cuda and t.is_floating_point() and t.numel():
                t.fill_(float("nan"))
            return t

        torch.empty = lambda *a, **k: poisoned(real_empty(*a, **k))
        torch.empty_like = lambda *a, **k: poisoned(real_empty_like(*a, **k))


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "stages.npz"))


@pytest.fixture(scope="session")
def charm_golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "charm.npz"))


@pytest.fixture(scope="session")
def synth_sd(manifest):
    """Full synthetic state dict: reference-manifest tensors + the CompressAI-side ones."""
    from dc_vic_amd.synth import full_synth_state_dict
    return full_synth_state_dict(seed=1234)


@pytest.fixture(scope="session")
def oracle(synth_sd):
    from oracle.dcvic_oracle import Oracle
    return Oracle(synth_sd)


@pytest.fixture(scope="session")
def oracle_compress(oracle):
    """Oracle.compress results cached per (image key, quality) for the whole session (a 512x768 pass costs ~15 s of CPU)."""
    cache = {}

    def get(key, x, q):
        k = (key, q)
        if k not in cache:
            cache[k] = oracle.compress(x, q)
        return cache[k]
    return get


def demo_image(name):
    """The reference's demo_images/*.png (data fixtures) as the CLI loads them: ToTensor + Normalize(.5, .5), compress.py:57-60."""
    import numpy as np
    import torch
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(GOLDEN, "demo_images", name)).convert("RGB"), dtype=np.uint8)
    x = torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255.0)
    return ((x - 0.5) / 0.5).unsqueeze(0)


@pytest.fixture(scope="session")
def train_golden():
    """tests/golden/train.npz: PatchGAN logits + the stage-3 loss values from the reference's own modules (oracle/gen_golden.py:gen_train)."""
    import numpy as np
    return np.load(os.path.join(GOLDEN, "train.npz"))


def train_golden_disc_state(G):
    """The discriminator weights the fixture was generated with: regenerated from its key -> shape manifest and seed."""
    from dc_vic_amd.synth import synth_discriminator_state
    return synth_discriminator_state(json.loads(str(G["d_manifest"])), int(G["d_seed"]))
