#!/usr/bin/env python3
"""Bulk decoding of compressed BLS12-377 points (decompress_bls12_377_g1/_g2_dev), wire bytes resident in HBM: kernel time
from HIP events, points/s, algorithmic GB/s (48 B in + 96 B out per G1 point, 96 B + 192 B per G2 point) against the 8 TB/s
HBM roofline - this is integer-VALU work (one square root + the endomorphism subgroup test, 64-bit ladders, per point), the HBM fraction is reported
because it is the roofline the brief names.  Beside it: the oracle's C restatement on the host cores (bounded sample)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi
from oracle import cpu_oracle as co
from oracle.py import ecc
ffi.init(0)
out = {}
sizes = [int(a) for a in sys.argv[1:]] or [10, 14, 16, 18, 20]
for group, curve, gen, size, words in (("g1", ecc.E1_377, ecc.G1_377, 48, 12), ("g2", ecc.E2_377, ecc.G2_377, 96, 24)):
    # 256 distinct valid encodings, tiled (the work per point does not depend on the point)
    P, enc = gen, []
    for i in range(256):
        P = curve.add(curve.add(P, P), gen)
        enc.append(ecc.ser_point(curve, P if i % 2 else curve.neg(P)))
    tile = np.frombuffer(b"".join(enc), dtype=np.uint8)
    for log_n in sizes:
        n = 1 << log_n
        host = np.tile(tile, (n + 255) // 256)[: n * size]
        d_in = torch.from_numpy(host.copy()).cuda()
        d_out = torch.zeros((n, words), dtype=torch.int64, device="cuda")
        d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
        res = {}
        for check in (1, 0):
            best = None
            for _ in range(3):
                ffi.decompress_dev(group, d_in.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), bool(check))
                ms = ffi.decompress_last_ms()
                best = ms if best is None or ms < best else best
            assert not d_st.any().item()
            key = "checked" if check else "unchecked"
            res[key + "_ms"] = best
            res[key + "_points_per_s"] = n / (best * 1e-3)
            res[key + "_alg_GBps"] = n * (size + words * 8) / (best * 1e-3) / 1e9
            res[key + "_hbm_roofline_frac"] = res[key + "_alg_GBps"] / 8000.0
        if log_n == sizes[0]:
            T = co.lib().orc_hardware_threads()
            m = min(n, 256 if group == "g2" else 1024)
            res["cpu_port_1core_checked_points_per_s"] = m / co.time_decompress(group, host[: m * size].tobytes(), True, 1)
            m2 = min(n, 16 * T)
            res["cpu_port_allcores_checked_points_per_s"] = m2 / co.time_decompress(group, host[: m2 * size].tobytes(), True, T)
            res["cpu_threads"] = T
        out[f"{group}_2^{log_n}"] = res
# Jacobian -> affine (normalize_bls12_377_g1/_g2; host buffers, kernel time from HIP events): field arithmetic only, so random
# field elements serve as coordinates
rng = np.random.default_rng(3)
for group, words in (("g1", 6), ("g2", 12)):
    n = 1 << 20
    jac = rng.integers(0, 1 << 62, size=(n, 3 * words), dtype=np.int64).astype(np.uint64)
    jac[:, 5::6] &= np.uint64((1 << 56) - 1)
    best = None
    for _ in range(3):
        xy, inf = ffi.normalize(group, jac)
        ms = ffi.decompress_last_ms()
        best = ms if best is None or ms < best else best
    res = {"kernel_ms": best, "points_per_s": n / (best * 1e-3), "alg_GBps": n * 5 * words * 8 / (best * 1e-3) / 1e9}
    m = 1 << 14
    import time
    t0 = time.perf_counter(); co.normalize("g1_377" if group == "g1" else "g2_377", jac[:m]); res["cpu_port_1core_points_per_s"] = m / (time.perf_counter() - t0)
    out[f"normalize_{group}_2^20"] = res
print(json.dumps(out))
