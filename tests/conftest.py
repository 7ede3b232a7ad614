import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: test needs two or more GPUs in one box (left out of the collection otherwise)")


def pytest_collection_modifyitems(config, items):
    import torch

    # tests that put rank processes on different GPUs exist only where there are GPUs to put them on
    # (device_count() does not initialise the GPU)
    if torch.cuda.device_count() < 2:
        gone = [it for it in items if "multigpu" in it.keywords]
        if gone:
            items[:] = [it for it in items if "multigpu" not in it.keywords]
            config.hook.pytest_deselected(items=gone)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
