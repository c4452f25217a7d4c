"""pytest configuration: registers the ``gpu`` marker and shared helpers."""

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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the package no longer configures MIOpen at import; the GPU tests compare stock
    # fp32 convolutions with the float64 / CPU oracle and need the accurate solvers
    import pytorchhessianfree_amd

    pytorchhessianfree_amd.configure()
    # The CPU references of the GPU tests (stock models, autograd, the oracle PCG) run on the box's host cores; with
    # one thread per core of a 256-core host these small convolutions spend their time synchronising (bench.py's
    # cpu_baseline probes the thread count and lands on 16).  HF_TEST_CPU_THREADS=0: torch's default.
    cap = int(os.environ.get("HF_TEST_CPU_THREADS", "16"))
    if cap > 0 and torch.get_num_threads() > cap:
        torch.set_num_threads(cap)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
