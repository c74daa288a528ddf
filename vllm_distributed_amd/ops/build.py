"""Build the native libraries in-tree with hipcc for gfx950 (cross-compiles
without a GPU).

    python -m vllm_distributed_amd.ops.build [--force] [--jobs N]

Produces, next to this file:
    libvda_ops.so    HIP kernels of ops/csrc/*.hip (C ABI, loaded with ctypes)
    libvda_comm.so   peer-memory collectives of parallel/csrc
    libvda_host.so   host-only runtime of runtime/csrc (g++): shared-memory RPC ring
Object files and content hashes live in ops/build/ so unchanged sources are
not recompiled.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ARCH = "gfx950"

LIBS = {
    "libvda_ops.so": {
        "dir": os.path.join(HERE, "csrc"),
        "out": os.path.join(HERE, "libvda_ops.so"),
        "ldflags": [],
    },
    "libvda_comm.so": {
        "dir": os.path.join(PKG, "parallel", "csrc"),
        "out": os.path.join(PKG, "parallel", "libvda_comm.so"),
        "ldflags": ["-lpthread", "-lrt"],
    },
    # host-only runtime (no device code): shared-memory RPC ring, block hashing
    "libvda_host.so": {
        "dir": os.path.join(PKG, "runtime", "csrc"),
        "out": os.path.join(PKG, "runtime", "libvda_host.so"),
        "ldflags": ["-lpthread", "-lrt"],
        "host": True,
    },
}


# Per-file additions to the device flags.  The online-softmax kernels keep their MFMA accumulators in
# ordinary VGPRs: with the accumulator-register form the compiler copied every output register out and
# back around the conditional rescale (attention.hip: ~130 of ~470 vector instructions per key tile;
# mla.hip: 3296 accumulator moves in six kernels, 400 -> 269 registers).  Measured neutral on the
# memory-bound skinny GEMMs, which therefore keep the default.
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA_FLAGS = {f: _VGPR_FORM for f in ("attention.hip", "attention_prefill.hip", "mla.hip")}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def _digest(path: str, extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    d = os.path.dirname(path)
    for p in [path] + sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile_one(src: str, obj: str, flags: list[str], force: bool, cc: str | None = None) -> tuple[str, bool]:
    stamp = obj + ".sha"
    want = _digest(src, " ".join(flags))
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return src, False
    cmd = [cc or _hipcc(), *flags, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    return src, True


def build(force: bool = False, jobs: int | None = None, verbose: bool = True) -> list[str]:
    jobs = jobs or min(8, os.cpu_count() or 1)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
             "-fvisibility=hidden", "-Wno-unused-value", "-DNDEBUG"]
    built = []
    for name, spec in LIBS.items():
        srcdir = spec["dir"]
        if not os.path.isdir(srcdir):
            continue
        srcs = sorted(os.path.join(srcdir, f) for f in os.listdir(srcdir) if f.endswith((".hip", ".cpp")))
        if not srcs:
            continue
        objdir = os.path.join(HERE, "build", name.split(".")[0])
        os.makedirs(objdir, exist_ok=True)
        objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]
        changed = False
        host = spec.get("host", False)
        cc = (shutil.which("g++") or "g++") if host else None
        fl = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DNDEBUG"] if host else flags
        with ThreadPoolExecutor(jobs) as ex:
            for src, did in ex.map(lambda a: _compile_one(a[0], a[1], fl + ([] if host else EXTRA_FLAGS.get(
                    os.path.basename(a[0]), [])), force, cc), zip(srcs, objs)):
                changed |= did
                if verbose and did:
                    print(f"[build] compiled {os.path.relpath(src, PKG)}", flush=True)
        out = spec["out"]
        if changed or not os.path.exists(out):
            cmd = [cc, "-shared", "-fPIC", "-o", out, *objs, *spec["ldflags"]] if host else \
                [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs, *spec["ldflags"]]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed for {name}:\n{r.stdout}\n{r.stderr}")
            if verbose:
                print(f"[build] linked {os.path.relpath(out, PKG)}", flush=True)
        built.append(out)
    return built


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    for p in build(a.force, a.jobs):
        print(p)
