"""In-tree build of libresnmtf_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m resnmtf_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libresnmtf_hip.so")
SOURCES = [os.path.join(CSRC, "resnmtf_hip.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "resnmtf_kernels.hip.inc"), os.path.join(ROOT, "include", "resnmtf_hip.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", OUT] + SOURCES
    if verbose:
        print("[resnmtf_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
