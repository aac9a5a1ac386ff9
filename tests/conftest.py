import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFERENCE_DIR = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


def load_golden(name):
    data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: torch.from_numpy(data[k]) for k in data.files}


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(scope="session")
def lib_built():
    """Build (or reuse) libpointflow_hip.so once per session; hipcc cross-compiles without a GPU."""
    from pointmvsnet_amd import build
    return build.build(verbose=False)


@pytest.fixture(scope="session")
def dev():
    assert torch.cuda.is_available(), "gpu-marked test collected on a machine without a GPU"
    from pointmvsnet_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def report(name, **values):
    """Append measured parity errors to gpurun_out/parity_report.jsonl (merged back by gpurun) so the
    tolerances written in the tests can be audited against what the hardware actually produced."""
    import json
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(dict(name=name, **{k: float(v) for k, v in values.items()})) + "\n")
    except OSError:
        pass
