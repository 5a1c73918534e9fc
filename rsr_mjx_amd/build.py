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
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Three translation units from the one source: the cube kernels + the C ABI; the T-shape kernels and the Go2 kernels each on their own
# with the SLP vectoriser off (its packed-fp32 pairing costs the Go2 and T-shape kernels ~3 % and gains the cube kernels ~0.5 %;
# measured A/B on one box).  Kernels of one unit also perturb each other's register allocation: a unit holds one model family.
UNITS = [("rsr_main.o", []), ("rsr_tshape.o", ["-DRSR_TU_TSHAPE", "-fno-slp-vectorize"]), ("rsr_go2.o", ["-DRSR_TU_GO2", "-fno-slp-vectorize"])]


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS) or os.path.getmtime(__file__) > t


def compile_lib(lib: str = LIB, extra_flags=(), verbose: bool = False) -> str:
    """hipcc -c of every unit (in parallel), then the link; `extra_flags` go to every unit (e.g. -DRSR_PROFILE)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    stem = os.path.splitext(os.path.basename(lib))[0]
    objs, procs = [], []
    for obj, flags in UNITS:
        o = os.path.join(CSRC, f"{stem}.{obj}")
        cmd = [hipcc] + HIPCC_FLAGS + list(flags) + list(extra_flags) + ["-c", os.path.join(CSRC, SOURCES[0]), "-o", o]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        objs.append(o)
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, cwd=CSRC)
    for o in objs:
        os.remove(o)
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles the HIP extension for gfx950 (hipcc cross-compiles without a GPU).  Returns the .so path."""
    if not force and not _stale(LIB):
        return LIB
    return compile_lib(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
