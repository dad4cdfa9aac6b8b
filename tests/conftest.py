import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of oracle time (the remaining configs[4] shards); run with --run-slow or KT_SLOW=1")


def pytest_addoption(parser):
    parser.addoption("--run-slow", action="store_true", default=False, help="also run the tests marked slow")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--run-slow") or os.environ.get("KT_SLOW", "0") not in ("", "0"):
        return
    skip = pytest.mark.skip(reason="slow: run with --run-slow or KT_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure)."""
    from oracle import kt_oracle
    kt_oracle.lib()
    return kt_oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """On a GPU box, let torch initialise its HIP context BEFORE the engine's library makes its first HIP call: the tests
    that hand torch tensors to the engine (test_sharded_gpu, the two-GPU exchange) otherwise initialise torch after dozens
    of engines have come and gone in the same process, and once (round 4, call r04c) that late initialisation answered
    "No HIP GPUs are available" although the engine's own kernels had just run.  A no-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
