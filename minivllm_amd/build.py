"""Builds the native libraries in-tree with the ROCm toolchain.

* ``_hip_kernels.so``  - every gfx950 kernel under ``csrc/*.hip`` (hipcc, one object per
  source, C ABI launchers that take raw device pointers and a ``hipStream_t``).
* ``_host_core.so``    - host-side C++ (csrc/host/*.cpp), no GPU code.

``python -m minivllm_amd.build`` builds whatever is out of date; ``--force`` rebuilds all.
The libraries are loaded with ctypes, so no Python headers or torch headers are involved
and hipcc cross-compiles on a machine without a GPU.
"""
from __future__ import annotations

import argparse
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "csrc", "build")
HIP_LIB = os.path.join(HERE, "_hip_kernels.so")
HOST_LIB = os.path.join(HERE, "_host_core.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"command failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")


def _file_flags(src: str) -> list:
    """Extra compiler flags a source asks for: a line ``// hipcc-flags: ...`` among its first 40."""
    with open(src) as f:
        for _, line in zip(range(40), f):
            if line.startswith("// hipcc-flags:"):
                return line.split(":", 1)[1].split()
    return []


def build_hip(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    if not force and _newer(HIP_LIB, srcs + hdrs):
        return HIP_LIB                       # library is newer than every source: keep it
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall",
             "-Wno-unused-function", "-Wno-unused-variable"]
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(BUILD, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            jobs.append([hipcc] + flags + _file_flags(s) + ["-c", s, "-o", o])
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} HIP source(s) for {ARCH}", flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MAX_JOBS", 6)))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not _newer(HIP_LIB, objs):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
        if verbose:
            print(f"[build] linked {HIP_LIB}", flush=True)
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "host", "*.cpp")))
    if not srcs:
        return ""
    hdrs = sorted(glob.glob(os.path.join(CSRC, "host", "*.h")))
    if force or not _newer(HOST_LIB, srcs + hdrs):
        cxx = shutil.which("g++") or shutil.which("c++")
        _run([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB] + srcs)
        if verbose:
            print(f"[build] linked {HOST_LIB}", flush=True)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = True) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    build_all(args.force)
