#!/usr/bin/env python3
"""Run a script of this repo against an alternative build of libfmk_hip.so (A/B timing of a kernel change on one box):
    python tools/ab_lib.py finmlkit_amd/lib/ab/libfmk_hip_prev.so bench.py --no-extras --cpu-sample 0 --steps 20"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finmlkit_amd._ffi as ffi
ffi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
