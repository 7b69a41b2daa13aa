#!/usr/bin/env python3
"""Diagnostic: build the library with other compiler flags per translation unit, for A/B runs through tools/run_with_lib.py:
    python tools/build_variant.py <tag> "<flags of the main unit>" "<flags of the k <= 16 pass unit>"
-> tools/micro/libresnmtf_<tag>.so   (empty string = none; the product's own flags: resnmtf_amd/build.py)"""
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import build as B  # noqa: E402

tag, main_flags, k16_flags = sys.argv[1], shlex.split(sys.argv[2]), shlex.split(sys.argv[3])
out = os.path.join(ROOT, "tools", "micro", f"libresnmtf_{tag}.so")
common = [B.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC]
o_main, o_k16 = f"/tmp/variant_{tag}_main.o", f"/tmp/variant_{tag}_k16.o"
procs = [subprocess.Popen(common + main_flags + ["-DRESNMTF_SPLIT_TU", "-c", B.SOURCES[0], "-o", o_main]),
         subprocess.Popen(common + k16_flags + ["-c", B.PASS_K16, "-o", o_k16])]
if any(p.wait() != 0 for p in procs):
    raise SystemExit(f"{tag}: compile failed")
subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, o_main, o_k16], check=True)
print("built", out)
