#!/usr/bin/env python3
"""Round-3 open finding: k_gen_points<Fq2> (synthetic G2 bases: 64-bit double-and-add per lane) of the library in build/ -> .npy, or compare
with a saved array: which points differ, and whether each side's points are on the twist.  usage: r4_gen_compare.py save|cmp FILE [log_n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn
from oracle.py import ecc
from oracle import cpu_oracle as co
ffi.init(0)
mode, path = sys.argv[1], sys.argv[2]
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 17)
pts = syn.device_points("bls12_377_g2", n, 0x5EED2000 + 97 + 17).cpu().numpy().view(np.uint64).reshape(n, 24)
if mode == "save":
    np.save(path, pts)
    print("saved", pts.shape)
else:
    ref = np.load(path)
    bad = np.nonzero((ref != pts).any(axis=1))[0]
    print("points that differ from the main library's:", len(bad), "of", n, "first:", bad[:8].tolist())
    for i in bad[:3]:
        for name, arr in (("main", ref), ("variant", pts)):
            v = co.from_mont(arr[i].reshape(4, 6), ecc.Q377)
            P = ((v[0], v[1]), (v[2], v[3]))
            print("  point", int(i), name, "on curve:", ecc.E2_377.on_curve(P))
