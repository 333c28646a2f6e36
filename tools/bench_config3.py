#!/usr/bin/env python3
"""BASELINE config 3 shape: 4096 batches x 256 (pk in G2, sig in G1): per batch one G2 MSM + one G1 MSM with 136-bit
exponents (Batch::verify, crates/bls-crypto/src/bls/batch.rs:44-84) + a 2-pair product check.  Synthetic points
(k_i*G), random exponents: measures the kernels' throughput; verdict parity for this flow is in tests/test_batch_gpu.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, codec, bls

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
force_c = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ffi.init(0)
if force_c:
    ffi.set_window_bits("bls12_377_g1", force_c); ffi.set_window_bits("bls12_377_g2", force_c)
G1 = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
      241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
tot = m * n
g1gen, _ = codec.pack_affine([G1], codec.Q377)
g2gen, _ = codec.pack_affine([bls.G2_GENERATOR], codec.Q377, ext=2)
t1 = torch.empty(tot * 12, dtype=torch.int64, device="cuda"); ffi.gen_points_dev("bls12_377_g1", t1.data_ptr(), tot, 7, g1gen.reshape(-1))
t2 = torch.empty(tot * 24, dtype=torch.int64, device="cuda"); ffi.gen_points_dev("bls12_377_g2", t2.data_ptr(), tot, 8, g2gen.reshape(-1))
sigs = t1.cpu().numpy().view(np.uint64).reshape(tot, 12); pks = t2.cpu().numpy().view(np.uint64).reshape(tot, 24)
rng = np.random.default_rng(1)
sc = np.zeros((tot, 4), dtype=np.uint64)
sc[:, 0] = rng.integers(0, 1 << 63, size=tot, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
sc[:, 1] = rng.integers(0, 1 << 63, size=tot, dtype=np.int64).astype(np.uint64) * np.uint64(2)
sc[:, 2] = rng.integers(0, 256, size=tot, dtype=np.int64).astype(np.uint64)          # 136-bit exponents
offs = np.arange(0, tot + 1, n, dtype=np.uint32)
res = {}
for grp, pts in (("bls12_377_g1", sigs), ("bls12_377_g2", pks)):
    ffi.msm_batch(grp, pts, None, sc, offs)
    t0 = time.perf_counter(); out = ffi.msm_batch(grp, pts, None, sc, offs); dt = time.perf_counter() - t0
    tm = ffi.msm_timings(grp)
    res[grp] = {"wall_ms": dt * 1e3, "device_ms": tm["total_ms"], "accumulate_ms": tm["accumulate_ms"], "convert_ms": tm["convert_ms"], "windows": tm["windows"], "sort_ms": tm["sort_ms"], "reduce_ms": tm["reduce_ms"],
                "window_bits": tm["window_bits"], "scalar_muls_per_s_device": tot / (tm["total_ms"] * 1e-3)}
# pairing part: 2 pairs per batch (inputs: any valid points)
g1 = sigs[: 2 * m]; g2 = pks[: 2 * m]
po = np.arange(0, 2 * m + 1, 2, dtype=np.uint32)
ffi.pairing_product_is_one_batch(g1, None, g2, None, po)
t0 = time.perf_counter(); ffi.pairing_product_is_one_batch(g1, None, g2, None, po); dt = time.perf_counter() - t0
res["pairing"] = dict(ffi.pairing_timings(), wall_ms=dt * 1e3, miller_loops=2 * m, final_exps=m)
res["config"] = {"batches": m, "signers_per_batch": n, "exponent_bits": 136}
dev_total = res["bls12_377_g1"]["device_ms"] + res["bls12_377_g2"]["device_ms"] + res["pairing"]["total_ms"]
res["batches_verified_per_s_device"] = m / (dev_total * 1e-3)
print(json.dumps(res))
