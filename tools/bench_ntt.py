#!/usr/bin/env python3
"""NTT over Fr(BW6-761) (ntt_bw6_761_fr_dev), device-resident, sizes 2^16..2^24: kernel times from HIP events, algorithmic GB/s
(one 48-B read + one 48-B write per element) against the 8 TB/s HBM roofline, and the oracle's single-thread time at 2^16."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi
from oracle import cpu_oracle as co
from oracle.py import ntt as ontt, ecc
ffi.init(0)
out = {}
for log_n in [int(a) for a in sys.argv[1:]] or [16, 18, 20, 22, 24]:
    n = 1 << log_n
    w = co.to_mont([ontt.root_of_unity(log_n)], ecc.Q377)[0]
    x = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 6), dtype=np.int64)
    x[:, 5] &= (1 << 56) - 1
    d = torch.from_numpy(x).cuda()
    ffi.ntt_dev(d.data_ptr(), log_n, w)          # builds the twiddle table
    best = None
    for _ in range(5):
        ffi.ntt_dev(d.data_ptr(), log_n, w)
        tm = ffi.ntt_timings()
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = tm
    best["elements_per_s"] = n / (best["total_ms"] * 1e-3)
    best["alg_GBps"] = n * 96 / (best["total_ms"] * 1e-3) / 1e9
    best["hbm_roofline_frac"] = best["alg_GBps"] / 8000.0
    if log_n <= 18:
        secs = co.time_ntt_fq377(x.view(np.uint64), log_n, ontt.root_of_unity(log_n))
        best["cpu_port_1core_elements_per_s"] = n / secs
    out[log_n] = best
print(json.dumps(out))
