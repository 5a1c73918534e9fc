"""Builds librsrmjx.so (HIP, gfx950) in-tree next to its sources."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "librsrmjx.so")
SOURCES = ["rsr_mjx.hip"]
HEADERS = ["rsr_device.hpp", "rsr_solver.hpp", os.path.join("..", "..", "include", "rsr_mjx.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles the HIP extension for gfx950 (hipcc cross-compiles without a GPU).  Returns the .so path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
