#!/usr/bin/env python3
"""Device-resident MSM timings for every group (BASELINE configs 4/5 shapes at single-GPU sizes): BLS12-377 G1/G2 and
BW6-761 G1/G2.  Inputs generated on the device (P_i = k_i*G), uniform scalars below the group order; prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, codec, bls

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = 3
ffi.init(0)
G1 = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
      241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
vk = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_vectors.json")))["groth16_bw6_761"]["vk"]
vkb = bytes.fromhex(vk)
alpha = codec.decompress_bw6_761(vkb[0:96]); beta = codec.decompress_bw6_761(vkb[96:192], g2=True)
gens = {"bls12_377_g1": codec.pack_affine([G1], codec.Q377)[0], "bls12_377_g2": codec.pack_affine([bls.G2_GENERATOR], codec.Q377, ext=2)[0],
        "bw6_761_g1": codec.pack_affine([alpha], codec.Q761)[0], "bw6_761_g2": codec.pack_affine([beta], codec.Q761)[0]}
ALG = {"bls12_377_g1": 128, "bls12_377_g2": 224, "bw6_761_g1": 240, "bw6_761_g2": 240}
n = 1 << logn
out = {"n": n}
rng = np.random.default_rng(3)
for grp in ("bls12_377_g1", "bls12_377_g2", "bw6_761_g1", "bw6_761_g2"):
    A, S, O = ffi.GROUP_SHAPE[grp]
    t = torch.empty(n * A, dtype=torch.int64, device="cuda")
    ffi.gen_points_dev(grp, t.data_ptr(), n, 11, gens[grp].reshape(-1))
    sc = rng.integers(0, 1 << 63, size=(n, S), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, S), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, S - 1] &= np.uint64((1 << (60 if S == 4 else 56)) - 1)     # < 2^252 / < 2^376: below r
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    ffi.msm_dev(grp, t.data_ptr(), 0, d.data_ptr(), n)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); ffi.msm_dev(grp, t.data_ptr(), 0, d.data_ptr(), n); dt = time.perf_counter() - t0
        tm = ffi.msm_timings(grp)
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = dict(tm, wall_ms=dt * 1e3)
    best["scalar_muls_per_s"] = n / (best["total_ms"] * 1e-3)
    best["alg_GBps_total"] = n * ALG[grp] / (best["total_ms"] * 1e-3) / 1e9
    out[grp] = best
    del t, d
print(json.dumps(out))
