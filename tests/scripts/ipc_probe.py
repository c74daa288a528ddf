"""Times hipIpcOpenMemHandle for one region size (exporter / importer processes on one GPU).
usage: ipc_probe.py export|import <workdir> <size_mib> [touch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from vllm_distributed_amd.device_allocator import alloc_shareable, open_shared  # noqa: E402

role, work, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) << 20
extra = int(sys.argv[4]) << 20 if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
if role == "export":
    ballast = torch.zeros(extra, dtype=torch.uint8, device=dev) if extra else None
    r = alloc_shareable(n, dev)
    r.tensor[:16] = 5
    torch.cuda.synchronize()
    with open(f"{work}/h.tmp", "wb") as f:
        f.write(r.handle)
    os.replace(f"{work}/h.tmp", f"{work}/h")
    t0 = time.time()
    while not os.path.exists(f"{work}/done") and time.time() - t0 < 20:
        time.sleep(0.1)
else:
    import faulthandler
    faulthandler.dump_traceback_later(15, exit=True)
    ballast = torch.zeros(extra, dtype=torch.uint8, device=dev) if extra else None
    while not os.path.exists(f"{work}/h"):
        time.sleep(0.05)
    h = open(f"{work}/h", "rb").read()
    t = time.time()
    r = open_shared(h, n, dev)
    print(f"open {n >> 20} MiB (+{extra >> 20} MiB torch ballast): {(time.time() - t) * 1e3:.1f} ms ok={r.tensor[:2].tolist()}",
          flush=True)
    open(f"{work}/done", "w").close()
