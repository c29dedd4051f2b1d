#!/usr/bin/env python3
"""Kernel-tuning aid: run bench.py against an alternative build of the library.   usage: ab_bench.py <lib.so> [bench args...]
(The product loads transgo_amd/libtransgo_hip.so only; this script points the loader elsewhere for one process.)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transgo_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
