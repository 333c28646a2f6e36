#!/usr/bin/env python3
"""A/B helper: run a tool script against an alternative build of the library (celo-bls-snark-rs_amd/build/ab/libcelo_bls_amd.so).
usage: python tools/ab_run.py tools/bench_pairing.py 81920"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celo_bls_snark_rs_amd import ffi
ffi.LIB_PATH = os.path.join(os.path.dirname(ffi.LIB_PATH), "ab", "libcelo_bls_amd.so")
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
