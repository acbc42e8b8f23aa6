import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a box without a HIP device, so a plain `pytest tests` there shows real results"""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run on the GPU box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU oracle (test infrastructure) is compiled on demand."""
    from oracle import binding
    binding.build()


@pytest.fixture(autouse=True)
def _collect_garbage_between_tests():
    """Simulation handles own device memory; a handle that dies in a LATER test -- whenever the cyclic collector happens to run,
    e.g. inside that test's HIP-graph capture -- frees it there.  Collect after every test instead."""
    yield
    import gc
    gc.collect()
