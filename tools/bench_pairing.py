#!/usr/bin/env python3
"""Pairing throughput vs batch size (2-pair products, BLS12-377).  Inputs are arbitrary valid points (k_i*G1, k_j*G2)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, codec, bls
ffi.init(0)
G1 = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
      241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
g1gen, _ = codec.pack_affine([G1], codec.Q377); g2gen, _ = codec.pack_affine([bls.G2_GENERATOR], codec.Q377, ext=2)
res = {}
for m in [int(a) for a in sys.argv[1:]] or [2048, 8192, 32768]:
    k = 2 * m
    t1 = torch.empty(k * 12, dtype=torch.int64, device="cuda"); ffi.gen_points_dev("bls12_377_g1", t1.data_ptr(), k, 7, g1gen.reshape(-1))
    t2 = torch.empty(k * 24, dtype=torch.int64, device="cuda"); ffi.gen_points_dev("bls12_377_g2", t2.data_ptr(), k, 8, g2gen.reshape(-1))
    g1 = t1.cpu().numpy().view(np.uint64).reshape(k, 12); g2 = t2.cpu().numpy().view(np.uint64).reshape(k, 24)
    offs = np.arange(0, k + 1, 2, dtype=np.uint32)
    ffi.pairing_product_is_one_batch(g1, None, g2, None, offs)
    t0 = time.perf_counter(); ffi.pairing_product_is_one_batch(g1, None, g2, None, offs); dt = time.perf_counter() - t0
    tm = ffi.pairing_timings()
    res[m] = dict(tm, wall_ms=dt * 1e3, miller_loops_per_s=k / (tm["total_ms"] * 1e-3), products_per_s=m / (tm["total_ms"] * 1e-3))
print(json.dumps(res))
