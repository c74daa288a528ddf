"""Platform layer: exactly two targets, MI355X (gfx950 through PyTorch-ROCm)
and a CPU plumbing path used by tests.

Reference: vllm/platforms/interface.py (Platform interface), vllm/platforms/rocm.py
(fp8 dtype :392-402, xGMI connectivity :259-281, custom all-reduce gate :409-414).
"""
from __future__ import annotations

import os
from functools import lru_cache

import torch


class Platform:
    name = "cpu"
    device_type = "cpu"
    dist_backend = "gloo"
    wave_size = 64

    def is_rocm(self) -> bool:
        return False

    def is_cpu(self) -> bool:
        return True

    def fp8_dtype(self) -> torch.dtype:
        return torch.float8_e4m3fn

    def supports_fp8(self) -> bool:
        return False

    def device_count(self) -> int:
        return 1

    def total_memory(self, device=0) -> int:
        import psutil
        return psutil.virtual_memory().total

    def mem_get_info(self, device=0) -> tuple[int, int]:
        import psutil
        vm = psutil.virtual_memory()
        return vm.available, vm.total

    def synchronize(self) -> None:
        pass

    def use_custom_allreduce(self) -> bool:
        return False

    def is_fully_connected(self, device_ids) -> bool:
        return False

    def check_and_update_config(self, cfg) -> None:
        cfg.compilation_config.level = 0

    def seed_everything(self, seed: int) -> None:
        import random
        import numpy as np
        random.seed(seed)
        np.random.seed(seed % (2**32))
        torch.manual_seed(seed)


class RocmPlatform(Platform):
    name = "rocm"
    device_type = "cuda"
    dist_backend = "nccl"   # RCCL under PyTorch-ROCm

    def is_rocm(self) -> bool:
        return True

    def is_cpu(self) -> bool:
        return False

    def supports_fp8(self) -> bool:
        return True

    def device_count(self) -> int:
        return torch.cuda.device_count()

    @lru_cache(maxsize=8)
    def gcn_arch(self, device=0) -> str:
        return torch.cuda.get_device_properties(device).gcnArchName.split(":")[0]

    def total_memory(self, device=0) -> int:
        return torch.cuda.get_device_properties(device).total_memory

    def mem_get_info(self, device=0) -> tuple[int, int]:
        return torch.cuda.mem_get_info(device)

    def synchronize(self) -> None:
        torch.cuda.synchronize()

    def use_custom_allreduce(self) -> bool:
        return True

    def is_fully_connected(self, device_ids) -> bool:
        """All pairs can map each other's memory (xGMI mesh inside one node)."""
        ids = list(device_ids)
        for i in ids:
            for j in ids:
                if i != j and not torch.cuda.can_device_access_peer(i, j):
                    return False
        return True

    def check_and_update_config(self, cfg) -> None:
        arch = self.gcn_arch(0)
        if arch != "gfx950" and not os.environ.get("VDA_ALLOW_OTHER_ARCH"):
            raise RuntimeError(f"this build targets gfx950 (MI355X) only, found {arch}")

    def seed_everything(self, seed: int) -> None:
        super().seed_everything(seed)
        torch.cuda.manual_seed_all(seed)


@lru_cache(maxsize=1)
def _detect() -> Platform:
    if torch.cuda.is_available() and getattr(torch.version, "hip", None):
        return RocmPlatform()
    return Platform()


class _Lazy:
    def __getattr__(self, k):
        return getattr(_detect(), k)


current_platform: Platform = _Lazy()  # type: ignore[assignment]
