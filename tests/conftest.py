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
    # The CPU references of the GPU tests (stock models, autograd, the oracle PCG) run on the box's host cores with
    # torch's default thread count.  HF_TEST_CPU_THREADS=16 makes them ~3.5x faster on a 256-core host (bench.py's
    # cpu_baseline probe lands on 16 threads too) but moves the fp32 CPU results by their own rounding (other reduction
    # partitions): the tightest comparisons are calibrated to the default (m_1 of the ResNet-18 solve: 1.04e-5 against
    # its 1e-5 bound with 16 threads) -- an option for local runs, not the default.
    cap = int(os.environ.get("HF_TEST_CPU_THREADS", "0"))
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
