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
    config.addinivalue_line("markers", "isolated: on hardware, run the test body in a child pytest process (device code "
                                       "that has never run on an MI355X: a GPU fault there must not end this session)")


def _isolate(item):
    if item.get_closest_marker("isolated") is None or os.environ.get("PF_TEST_INNER") == "1":
        return False
    return torch.cuda.is_available() or os.environ.get("PF_TEST_ISOLATE") == "force"      # ("force": the mechanism's own test)


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """``@pytest.mark.isolated``: the test runs as ``python -m pytest <nodeid> --runxfail`` in a child process and passes
    iff that exits 0.  A memory fault or an abort inside a never-yet-run kernel kills the child; this session, with the
    results of every test before it, goes on.  (The child appends its own lines to gpurun_out/parity_report.jsonl.)"""
    if not _isolate(pyfuncitem):
        return None
    import subprocess
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-m", "gpu", "-q", "-x", "--runxfail", "-p", "no:cacheprovider"]
    env = dict(os.environ, PF_TEST_INNER="1")
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True,
                           timeout=float(os.environ.get("PF_TEST_INNER_TIMEOUT", "900")))
    except subprocess.TimeoutExpired as exc:
        out = exc.stdout if isinstance(exc.stdout, str) else (exc.stdout or b"").decode("utf8", "replace")
        pytest.fail("isolated test killed after %s s:\n%s" % (exc.timeout, out[-3000:]), pytrace=False)
    if r.returncode != 0:
        pytest.fail("isolated test failed in its child process (exit code %s):\n%s" % (r.returncode, r.stdout[-3000:]),
                    pytrace=False)
    return True


@pytest.fixture(autouse=True)
def _drop_in_switches_do_not_leak():
    """compat.install_as_pointmvsnet() turns on two PROCESS-WIDE switches of the drop-in route (per-module hipGraph replay,
    get_pixel_grids on the device).  A test that installs the route must not change what later tests of the session see:
    round 6's first whole-suite run failed test_frustum_variance_vs_reference_composition (it calls get_pixel_grids as the
    reference does and expects the reference's host tensor) only because a route test had run before it."""
    yield
    mods = sys.modules
    if "pointmvsnet_amd.graph" in mods:
        mods["pointmvsnet_amd.graph"].MODULE_GRAPHS = False
    if "pointmvsnet_amd.functions.functions" in mods:
        mods["pointmvsnet_amd.functions.functions"].PIXEL_GRID_ON_DEVICE = False


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
    """cuda:0 -- or, with PF_EMULATE=1 on a machine WITHOUT a GPU, the CPU with tests/hipemu behind the C ABI for the whole
    session: the gpu-marked tests then execute the kernel sources on the host (tests/hipemu/on_cpu.py).  A development
    aid for rounds without GPU access (slow, no timing, not a substitute: the driver's `pytest -m gpu` runs on an MI355X)."""
    if os.environ.get("PF_EMULATE") == "1" and not torch.cuda.is_available():
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        from on_cpu import emulated_gpu
        with emulated_gpu():
            yield torch.device("cpu")
        return
    assert torch.cuda.is_available(), "gpu-marked test collected on a machine without a GPU"
    from pointmvsnet_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    yield torch.device("cuda:0")


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
