#!/usr/bin/env python3
"""One resident 2^k-term G1 MSM: a single call against the same job cut into s index ranges run side by side on the same device
(msm_*_multi_dev with the device listed s times: s engines, s streams) - does one shard's latency-bound tail hide under the
other's accumulation?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec
ffi.init(0)
out = {}
for log_n in [int(a) for a in sys.argv[1:]] or [20, 22]:
    n = 1 << log_n
    pts = syn.device_points("bls12_377_g1", n, 11)
    sc = syn.uniform_scalars("bls12_377_g1", n, 12)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    res = {}
    for s in (1, 2, 3, 4):
        per = n // s
        bases = [pts.data_ptr() + i * per * 96 for i in range(s)]
        scal = [d_sc.data_ptr() + i * per * 32 for i in range(s)]
        sizes = [per] * (s - 1) + [n - per * (s - 1)]
        run = (lambda: ffi.msm_dev("bls12_377_g1", pts.data_ptr(), 0, d_sc.data_ptr(), n)) if s == 1 else \
              (lambda: ffi.msm_multi_dev("bls12_377_g1", [0] * s, bases, None, scal, sizes))
        r = run(); run()
        t0 = time.perf_counter()
        for _ in range(20):
            r = run()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        from oracle.py import ecc
        res[f"shards_{s}"] = {"ms": ms, "smul_per_s": n / ms * 1e3, "x": hex(codec.jacobian_to_affine(r, ecc.Q377)[0])[:20]}
    out[f"2^{log_n}"] = res
print(json.dumps(out))
