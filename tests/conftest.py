import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# in-process engine core by default; the multi-process tests opt in explicitly
os.environ.setdefault("VLLM_ENABLE_V1_MULTIPROCESSING", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
