#!/usr/bin/env python3
"""msm_bls12_377_g1_subgroup_dev (GLV split) beside msm_bls12_377_g1_dev at several sizes and window sizes; device-resident inputs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn
ffi.init(0)
res = {}
GROUP = "bls12_377_g1"
args = sys.argv[1:]
if args and args[0] in ("g1", "g2"):
    GROUP = "bls12_377_" + args[0]
    args = args[1:]
for logn in [int(a) for a in args] or [14, 16, 17, 18, 20]:
    n = 1 << logn
    b = syn.device_points(GROUP, n, 5)
    sc = syn.uniform_scalars(GROUP, n, 6)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    row = {}
    for name, sub, cs in (("plain", False, [0]), ("subgroup", True, [0, 15, 16])):
        for c in cs:
            ffi.set_window_bits(GROUP, c)
            try:
                ffi.msm_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, subgroup=sub)
            except RuntimeError:
                continue
            best, wall = None, None
            for _ in range(5):
                t0 = time.perf_counter(); ffi.msm_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, subgroup=sub); dt = (time.perf_counter() - t0) * 1e3
                tm = ffi.msm_timings(GROUP)
                if best is None or tm["total_ms"] < best["total_ms"]:
                    best, wall = tm, dt
            row["%s_c%d" % (name, c)] = {"wall_ms": round(wall, 3), "dev_ms": round(best["total_ms"], 3), "acc": round(best["accumulate_ms"], 3), "red": round(best["reduce_ms"], 3),
                                         "conv": round(best["convert_ms"], 3), "sort": round(best["sort_ms"], 3), "c": best["window_bits"], "nw": best["windows"]}
    ffi.set_window_bits(GROUP, 0)
    res[logn] = row
    print(logn, json.dumps(row), flush=True)
