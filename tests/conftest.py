import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def rel_err(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="session")
def golden():
    return {n[:-4]: load_golden(n) for n in os.listdir(GOLDEN) if n.endswith(".npz")}


@pytest.fixture(scope="session")
def seeded_diffuser():
    """Drop-in GaussianDiffusion + Denoiser with the fixture weights (seed 0 + perturbed norms/biases)."""
    from posediffusion_amd import synth
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    return diff


@pytest.fixture(scope="session")
def oracle_weights(seeded_diffuser, golden):
    from oracle import pd_oracle as O
    from oracle.make_golden import weight_checksum
    sd = {k: v.detach().cpu() for k, v in seeded_diffuser.model.state_dict().items()}   # may live on the GPU by now
    np.testing.assert_allclose(weight_checksum(sd), golden["denoiser"]["weight_checksum"], rtol=1e-9,
                               err_msg="seeded weights differ from the ones the golden vectors were made with")
    return O.cast_state_dict(sd, torch.float32)


@pytest.fixture(scope="session")
def engine(seeded_diffuser):
    """A PoseEngine on cuda:0 built through the drop-in modules (GPU tests only)."""
    from posediffusion_amd.host import get_engine
    dev = torch.device("cuda:0")
    diff = seeded_diffuser.to(dev)
    eng = get_engine(diff.model, diff, 8, 50)
    yield eng
    seeded_diffuser.to("cpu")
