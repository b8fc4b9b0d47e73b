"""In-tree build of libtsff.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "tsff_kernels.hip")
DEPS = [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))] + [
    os.path.join(ROOT, "include", "tsff.h")]  # one translation unit: tsff_kernels.hip includes every .h / .inc of csrc/
OUT = os.path.join(_HERE, "libtsff.so")


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
