"""Builds a variant of the library next to the default one: python tools/build_variant.py <tag> [--rev GITREV] [-Dflags ...]
-> rsr_mjx_amd/csrc/librsrmjx_<tag>.so (for tools/ab_bench.py).  --rev builds the sources of another commit."""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
args = sys.argv[2:]
rev = None
if "--rev" in args:
    i = args.index("--rev"); rev = args[i + 1]; args = args[:i] + args[i + 2:]
from rsr_mjx_amd import build as B
out = os.path.join(B.CSRC, f"librsrmjx_{tag}.so")
if rev is None:
    B.compile_lib(out, extra_flags=args)
else:
    tmp = tempfile.mkdtemp()
    try:
        for f in ("rsr_mjx_amd/csrc/rsr_mjx.hip", "rsr_mjx_amd/csrc/rsr_device.hpp", "rsr_mjx_amd/csrc/rsr_solver.hpp", "include/rsr_mjx.h"):
            dst = os.path.join(tmp, f); os.makedirs(os.path.dirname(dst), exist_ok=True)
            open(dst, "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{f}"]))
        csrc = os.path.join(tmp, "rsr_mjx_amd", "csrc")
        old = B.CSRC; B.CSRC = csrc
        try:
            B.compile_lib(os.path.join(csrc, "lib.so"), extra_flags=args)
        finally:
            B.CSRC = old
        shutil.copy(os.path.join(csrc, "lib.so"), out)
    finally:
        shutil.rmtree(tmp)
print(out)
