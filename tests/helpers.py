"""Shared test helpers: seeded workloads and oracle/product comparisons."""
import numpy as np
from oracle.py import ecc
from oracle import cpu_oracle as co


def seeded_scalars(n, seed, modulus, edge=True):
    rng = ecc.SplitMix64(seed)
    sc = [ecc.random_scalar(rng, modulus) for _ in range(n)]
    if edge and n >= 8:
        sc[0] = 0
        sc[1] = 1
        sc[2] = modulus - 1
        sc[3] = 1 << 64
        sc[4] = (1 << 136) - 1  # Batch::verify-sized exponent (crates/bls-crypto/src/bls/batch.rs:23-28)
        sc[5] = 2
    return sc


def seeded_points(curve, gen, n, seed):
    rng = ecc.SplitMix64(seed)
    return [curve.mul(gen, rng.next() | 1) for _ in range(n)]


def scalars_np(sc, limbs):
    return co.ints_to_limbs(sc, limbs)


def splitmix64_at(seed, i):
    M = (1 << 64) - 1
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def build_hosttest():
    """Builds (make decides whether anything is stale) and returns the path of the host-only bounds-tracking library: host_test.cpp
    with -DCELO_FP_TRACK plus the host units it exercises (the IFMA Horner epilogue and the CPU probe)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "celo-bls-snark-rs_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "../build/libcelo_hosttest.so"])
    return os.path.join(root, "celo-bls-snark-rs_amd", "build", "libcelo_hosttest.so")
