"""Synthetic workloads of BASELINE.json's configurations, built on the device so that full-size
inputs need no host big-int loop.  Used by bench.py and the -m gpu tests; no oracle imports (the oracle checks, it does not
produce).

  cfg2  BLS12-377 G1 MSM: bases P_i = k_i * G (k_i = splitmix64(seed, i) | 1), uniform 252-bit scalars
  cfg3  Batch::verify (crates/bls-crypto/src/bls/batch.rs:44-84): m batches x n signers of VALID data -
        H_b = h_b * G1 stands for the message hash, sk_bj = splitmix64(seed, b n + j) | 1, pk_bj = sk_bj * g2,
        sig_bj = sk_bj * H_b; 136-bit exponents; a chosen set of batches is corrupted (one signature replaced)
  cfg4  BW6-761 G1 MSM: bases k_i * A (A = the alpha_g1 of the reference's own Groth16 test key, an r-torsion point:
        crates/bls-snark-sys/src/snark/mod.rs:52), uniform 376-bit scalars or a witness-like mix (about 60 % zeros and ones)
  cfg5  G1 + G2 MSMs and independent Miller loops, concurrently
"""
import numpy as np
import torch

from . import bls, codec, ffi

G1_GENERATOR = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
                241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
# alpha_g1 / beta_g2 of the Groth16 verifying key in crates/bls-snark-sys/src/snark/mod.rs:52 (decompressed): r-torsion points of
# BW6-761 G1 / G2, used as generators of synthetic bases (the reference tree holds no other BW6-761 point).
BW6_G1_POINT = (0xa8fcd52b7fa6f6edefc8c0bcba1c0f95045bcce706887b567eb5d6c5478bd21c7f995f9b8547737ce5cd2a484a7d5d96cee7e02e5bfa229f938bb0ec535ae3e3de957fe9e43eb2e418928872e5655d5092eae8b2bbdaf11ac60b349a40028a,
                0x699e25b0b93a51f5ea7ba40c97c50ffd73f14c77d3078feb989e4cedcdfb65cec88561c16ef31d9f96b220c86ba806c0506d574bb22e28b62a3d5c3f4c9dd11e078ead7cf9f05b20b886a2d04631c19e9a97d82695180f5955f9de371b0e7c)
BW6_G2_POINT = (0x2242a3e97f0b713cd64f7d1da9d8ee75c577b1e153b1cc153741956997322ffcb3dd329ec1591cf248b9f309d1059bc4f94785e9422b18b3fcfc79ee5457be6ebb95cb3136251a586c8c19345124cff4237a4f2f922809492e3867f2d70747,
                0xfc108d415344b37d0040426beb5af3c831222b725f6dcea587c11d5ccdebe397f462c1d48364b6cc3b82250391290bb7987c18c32a3bf96472989ada43fc32263c30e8b90f23314d3fce57d269dd515ae7600f7c749c14fb071083ba1ad65f)

ALG_BYTES = {"bls12_377_g1": 128, "bls12_377_g2": 224, "bw6_761_g1": 240, "bw6_761_g2": 240}   # SURVEY.md section 8d, per scalar-mul
ALG_BYTES_PER_MILLER_LOOP = 288


def generator_limbs(group):
    if group == "bls12_377_g1":
        return codec.pack_affine([G1_GENERATOR], codec.Q377)[0].reshape(-1)
    if group == "bls12_377_g2":
        return codec.pack_affine([bls.G2_GENERATOR], codec.Q377, ext=2)[0].reshape(-1)
    if group == "bw6_761_g1":
        return codec.pack_affine([BW6_G1_POINT], codec.Q761)[0].reshape(-1)
    if group == "bw6_761_g2":
        return codec.pack_affine([BW6_G2_POINT], codec.Q761)[0].reshape(-1)
    raise KeyError(group)


def neg_g2_limbs():
    (x0, x1), (y0, y1) = bls.G2_GENERATOR
    q = codec.Q377
    return codec.pack_affine([((x0, x1), (-y0 % q, -y1 % q))], codec.Q377, ext=2)[0].reshape(-1)


def device_points(group, n, seed, device="cuda"):
    """n affine points k_i * G of `group` in HBM (arkworks layout), as an int64 tensor of n * A words."""
    A = ffi.GROUP_SHAPE[group][0]
    t = torch.empty(n * A, dtype=torch.int64, device=device)
    ffi.gen_points_dev(group, t.data_ptr(), n, seed, generator_limbs(group))
    return t


def uniform_scalars(group, n, seed):
    """Uniform scalars below the group order (252 / 376 random bits), canonical limbs, numpy (n, S) uint64."""
    S = ffi.GROUP_SHAPE[group][1]
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 1 << 63, size=(n, S), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, S), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, S - 1] &= np.uint64((1 << (60 if S == 4 else 56)) - 1)
    return sc


def witness_like_scalars(group, n, seed):
    """The shape of a Groth16 witness: about 40 % zeros, 20 % ones, the rest uniform."""
    sc = uniform_scalars(group, n, seed)
    u = np.random.default_rng(seed ^ 0xABCDEF).random(n)
    sc[u < 0.4] = 0
    one = (u >= 0.4) & (u < 0.6)
    sc[one] = 0
    sc[one, 0] = 1
    return sc


def batch_exponents(tot, seed, bits=136):
    """`bits`-bit batching exponents in 4-limb containers (Batch::verify draws 128 + ceil(log2 n) random bits, batch.rs:23-28)."""
    rng = np.random.default_rng(seed)
    sc = np.zeros((tot, 4), dtype=np.uint64)
    raw = rng.integers(0, 256, size=(tot, (bits + 7) // 8), dtype=np.uint8)
    if bits % 8:
        raw[:, -1] &= (1 << (bits % 8)) - 1
    pad = np.zeros((tot, 32), dtype=np.uint8)
    pad[:, : raw.shape[1]] = raw
    sc[:] = pad.view(np.uint64).reshape(tot, 4)
    return sc


def valid_batches(m, n, seed, corrupt=(), device="cuda"):
    """cfg3 workload in HBM: returns dict(pk, sig, hash: int64 device tensors; offsets: uint32 numpy (m + 1); expect: uint8 (m)).
    Batches listed in `corrupt` get their signature 0 replaced by signature 1 (the batch then fails unless its exponents
    collide, probability 2^-136)."""
    tot = m * n
    hashes = device_points("bls12_377_g1", m, seed + 1, device)                 # H_b = h_b * G1
    h_host = hashes.cpu().numpy().view(np.uint64).reshape(m, 12)
    sig = torch.empty(tot * 12, dtype=torch.int64, device=device)
    ffi.gen_points_grouped_dev("bls12_377_g1", sig.data_ptr(), tot, seed + 2, h_host, n)          # sk_bj * H_b
    pk = torch.empty(tot * 24, dtype=torch.int64, device=device)
    ffi.gen_points_dev("bls12_377_g2", pk.data_ptr(), tot, seed + 2, generator_limbs("bls12_377_g2"))   # sk_bj * g2, same sk
    expect = np.ones(m, dtype=np.uint8)
    if len(corrupt):
        s2 = sig.view(tot, 12)
        idx = torch.as_tensor(np.asarray(corrupt, dtype=np.int64) * n, device=device)
        s2[idx] = s2[idx + 1]
        expect[np.asarray(corrupt, dtype=np.int64)] = 0
    torch.cuda.synchronize()
    return {"pk": pk, "sig": sig, "hash": hashes, "offsets": np.arange(0, tot + 1, n, dtype=np.uint32), "expect": expect, "m": m, "n": n}


def verify_products(m, seed, bad=None, device="cuda"):
    """m DISTINCT two-pair products e(sig_b, -g2) * e(H_b, pk_b) (PublicKey::verify / Batch::verify's final check, public.rs:102): every product
    has its own message point H_b = h_b * G1, its own key sk_b (pk_b = sk_b * g2, sig_b = sk_b * H_b), all generated on the device.  The
    products listed in `bad` (default: every 97th from 1) carry the NEXT product's signature, so they must be rejected.
    Returns host arrays (g1 (2 m, 12), g2 (2 m, 24), offsets (m + 1) uint32, expected accept list)."""
    if bad is None:
        bad = list(range(1, m - 1, 97))
    w = valid_batches(m, 1, seed, [b for b in bad if b < m - 1], device)
    sig = w["sig"].view(m, 12).cpu().numpy().view(np.uint64)
    hh = w["hash"].view(m, 12).cpu().numpy().view(np.uint64)
    pk = w["pk"].view(m, 24).cpu().numpy().view(np.uint64)
    g1 = np.empty((2 * m, 12), dtype=np.uint64)
    g2 = np.empty((2 * m, 24), dtype=np.uint64)
    g1[0::2] = sig; g1[1::2] = hh
    g2[0::2] = neg_g2_limbs(); g2[1::2] = pk
    return g1, g2, np.arange(0, 2 * m + 1, 2, dtype=np.uint32), [int(x) for x in w["expect"]]


def random_pair_products(m, k, seed, device="cuda"):
    """m DISTINCT products of k pairs of UNRELATED points (P_i = a_i * G1, Q_i = b_i * g2, all different): no shared first G2 point, the product
    is not 1.  Returns host arrays (g1 (k m, 12), g2 (k m, 24), offsets (m + 1) uint32)."""
    g1 = device_points("bls12_377_g1", k * m, seed, device).view(k * m, 12).cpu().numpy().view(np.uint64)
    g2 = device_points("bls12_377_g2", k * m, seed + 1, device).view(k * m, 24).cpu().numpy().view(np.uint64)
    return g1, g2, np.arange(0, k * m + 1, k, dtype=np.uint32)
