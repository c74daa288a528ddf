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


# ``VDA_TEST_DEVICE=cuda`` re-runs the CPU model tests with every engine on the GPU (fp32, PyTorch reference ops through
# ``VDA_FORCE_TORCH_OPS=1``): the same token-exact assertions then check the models' own device handling -- position
# tables, masks and index tensors built inside towers and processors -- which the CPU runs cannot see.
if os.environ.get("VDA_TEST_DEVICE"):
    import vllm_distributed_amd as _pkg

    _init = _pkg.LLM.__init__

    def _on_device(self, *a, **kw):
        if kw.get("device") == "cpu":
            kw["device"] = os.environ["VDA_TEST_DEVICE"]
            kw.setdefault("gpu_memory_utilization", 0.05)
            kw.setdefault("enforce_eager", True)
        return _init(self, *a, **kw)

    _pkg.LLM.__init__ = _on_device
