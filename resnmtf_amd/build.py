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
# the k <= 16 streaming pass: a translation unit of its own, compiled with the max-ILP machine scheduler (csrc/resnmtf_split_tu.h)
PASS_K16 = os.path.join(CSRC, "resnmtf_pass_k16.hip")
PASS_K16_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
DEPS = SOURCES + [PASS_K16, os.path.join(CSRC, "resnmtf_kernels.hip.inc"), os.path.join(CSRC, "resnmtf_split_tu.h"),
                  os.path.join(ROOT, "include", "resnmtf_hip.h")]


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
    common = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs = [os.path.join(HERE, "_build_" + os.path.basename(src) + ".o") for src in SOURCES + [PASS_K16]]
    cmds = [common + ["-DRESNMTF_SPLIT_TU", "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
    cmds.append(common + PASS_K16_FLAGS + ["-c", PASS_K16, "-o", objs[-1]])
    if verbose:
        for cmd in cmds:
            print("[resnmtf_amd.build]", " ".join(cmd), flush=True)
    procs = [subprocess.Popen(cmd) for cmd in cmds]      # (the two units side by side)
    if any(p.wait() != 0 for p in procs):
        raise subprocess.CalledProcessError(1, cmds[0])
    link = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUT] + objs
    if verbose:
        print("[resnmtf_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    for obj in objs:
        os.remove(obj)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
