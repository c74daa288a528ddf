"""Allocates from the sleep allocator and exits normally: the exit status shows whether process
teardown with live pools is clean."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vllm_distributed_amd.device_allocator import SleepAllocator

a = SleepAllocator.get()
with a.use_memory_pool("weights"):
    w = torch.ones(1 << 24, device="cuda:0")
a.sleep()
a.wake_up()
print("sum", float(w.sum()), flush=True)
print("probe done", flush=True)
