#!/usr/bin/env python3
"""Diagnostic: run one of the tools against another build of the library (A/B comparisons of compile-time variants):
    python tools/run_with_lib.py tools/micro/libresnmtf_d4.so tools/bench_configs.py c5v1"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resnmtf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
