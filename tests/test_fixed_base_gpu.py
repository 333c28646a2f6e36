"""GPU (-m gpu): the fixed-base MSM (include/celo_bls_amd.h msm_*_precompute / msm_*_fixed) - per-key tables for the Groth16 prover's
queries, whose Parameters stay while the assignment changes (crates/epoch-snark/src/api/prover.rs:78,112; setup.rs:63-105).
Same group element as VariableBaseMSM on the same inputs: checked against the oracle (C++ port of ark-ec's Pippenger, Python big-int
definition at small n) and against the variable-base entry point, for every table window size, all four groups, edge scalars, flagged
identities, a base of order 2 (its 2^(c j) multiples ARE the identity), fewer scalars than bases, witness-like scalars, several calls
on one handle and two handles alive at once.  Integer work: bit-exact (affine-normalised)."""
import os
import threading
import numpy as np
import pytest
import torch
from oracle.py import ecc
from oracle import cpu_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu
KIND = {"bls12_377_g1": "g1_377", "bls12_377_g2": "g2_377", "bw6_761_g1": "761", "bw6_761_g2": "761"}
AFF = {"bls12_377_g1": 12, "bls12_377_g2": 24, "bw6_761_g1": 24, "bw6_761_g2": 24}


def _threads():
    return max(2, min(32, os.cpu_count() or 2))


@pytest.mark.parametrize("n", [1, 2, 33, 300])
def test_g1_small_vs_python_every_window_size(gpu, n):
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 4100 + n)
    sc = H.seeded_scalars(n, 4200 + n, ecc.R377)
    if n >= 33:
        pts[7] = None                                   # flagged identity
        pts[9] = pts[8]; sc[9] = sc[8]                  # equal bases, equal scalars: the doubling branch inside a bucket
        pts[11] = ecc.E1_377.neg(pts[10]); sc[11] = sc[10]      # opposite bases: cancellation
        pts[12] = (ecc.Q377 - 1, 0); sc[12] = (1 << 200) + 1    # a point of order 2: every 2^(c j) multiple with j >= 1 is the identity
        sc[13] = (1 << 253) - 1 - (1 << 17)             # every top digit set (bits above Fr::MODULUS_BITS are not part of the scalar)
    xy, inf = co.pack_g1_377(pts)
    s = H.scalars_np(sc, 4)
    want = ecc.E1_377.msm(pts, [k & ((1 << 253) - 1) for k in sc])
    for cf in (0, 16, 17, 18, 19, 20):
        fb = gpu.FixedBase("bls12_377_g1", xy, inf, window_bits=cf)
        info = fb.info()
        assert info["n"] == n and info["window_bits"] == (cf or 16) and info["windows"] == (253 + info["window_bits"]) // info["window_bits"]
        assert co.jac_to_affine(fb.msm(s), "g1_377") == want, (n, cf)
        fb.release()


@pytest.mark.parametrize("group,n,cfs", [("bls12_377_g1", 1 << 15, [0, 16, 20]), ("bls12_377_g1", 1 << 18, [0, 21]), ("bls12_377_g2", 3000, [16, 18]),
                                         ("bls12_377_g2", 1 << 16, [0]), ("bw6_761_g1", 3000, [16, 19]), ("bw6_761_g1", 1 << 17, [0, 20]), ("bw6_761_g2", 1 << 15, [17])])
def test_fixed_equals_oracle_and_variable_base(gpu, group, n, cfs):
    from celo_bls_snark_rs_amd import synthetic as syn
    pts = syn.device_points(group, n, 4300 + n)
    sc = syn.uniform_scalars(group, n, 4301 + n)
    S = sc.shape[1]
    r = ecc.R377 if group.startswith("bls") else ecc.R761
    sc[:5] = H.scalars_np([0, 1, r - 1, 1 << 64, 2], S)
    h = pts.cpu().numpy().view(np.uint64).reshape(n, AFF[group])
    inf = np.zeros(n, dtype=np.uint8); inf[5] = 1
    d_inf = torch.from_numpy(inf).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    exp = co.jac_to_affine(co.msm(group, h, inf, sc, threads=_threads()), KIND[group])
    assert co.jac_to_affine(gpu.msm_dev(group, pts.data_ptr(), d_inf.data_ptr(), d_sc.data_ptr(), n), KIND[group]) == exp
    for cf in cfs:
        fb = gpu.FixedBase(group, d_bases=pts.data_ptr(), d_inf=d_inf.data_ptr(), n=n, window_bits=cf)
        assert co.jac_to_affine(fb.msm_dev(d_sc.data_ptr(), n), KIND[group]) == exp, (group, n, cf)
        assert co.jac_to_affine(fb.msm(sc), KIND[group]) == exp                       # host scalars, same handle
        # fewer scalars than bases: the shorter side decides (VariableBaseMSM zips)
        k = n - n // 3
        exp_k = co.jac_to_affine(co.msm(group, h[:k], inf[:k], sc[:k], threads=_threads()), KIND[group])
        assert co.jac_to_affine(fb.msm_dev(d_sc.data_ptr(), k), KIND[group]) == exp_k
        # a second scalar vector on the same tables: witness-like (about 60 % zeros and ones)
        sc2 = syn.witness_like_scalars(group, n, 4400 + n)
        exp2 = co.jac_to_affine(co.msm(group, h, inf, sc2, threads=_threads()), KIND[group])
        assert co.jac_to_affine(fb.msm(sc2), KIND[group]) == exp2
        fb.release()


def test_two_handles_and_concurrent_callers(gpu):
    """Two keys' tables alive at once, each used from two host threads at the same time (the prover's four MSMs are concurrent)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 20000
    keys = []
    for q, group in enumerate(("bw6_761_g1", "bls12_377_g1")):
        pts = syn.device_points(group, n, 4500 + q)
        h = pts.cpu().numpy().view(np.uint64).reshape(n, AFF[group])
        keys.append((group, h, gpu.FixedBase(group, h, None)))
    jobs, out, errs = [], {}, []
    for q, (group, h, fb) in enumerate(keys):
        for t in range(2):
            sc = syn.uniform_scalars(group, n, 4600 + 10 * q + t)
            jobs.append((q, t, group, h, fb, sc))

    def run(q, t, group, h, fb, sc):
        try:
            out[(q, t)] = co.jac_to_affine(fb.msm(sc), KIND[group])
        except Exception as e:                          # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=run, args=j) for j in jobs]
    for x in th: x.start()
    for x in th: x.join()
    assert not errs, errs
    for q, t, group, h, fb, sc in jobs:
        assert out[(q, t)] == co.jac_to_affine(co.msm(group, h, None, sc, threads=_threads()), KIND[group])
    for _, _, fb in keys:
        fb.release()


def test_bad_arguments_are_refused(gpu):
    xy, inf = co.pack_g1_377(H.seeded_points(ecc.E1_377, ecc.G1_377, 4, 1))
    with pytest.raises(RuntimeError):
        gpu.FixedBase("bls12_377_g1", xy, inf, window_bits=15)
    with pytest.raises(RuntimeError):
        gpu.FixedBase("bls12_377_g1", xy, inf, window_bits=23)
    fb = gpu.FixedBase("bls12_377_g1", xy, inf)
    assert co.jac_to_affine(fb.msm(np.zeros((0, 4), dtype=np.uint64)), "g1_377") is None        # no scalars: the identity
    fb.release()


def test_automatic_window_steps_down_for_a_large_key(gpu):
    """with window_bits = 0 a key too large for the preferred window (BW6-761 at c = 21: n W NV >= 2^32 from ~2^22.8 terms) used to be
    refused with rc = 2 at the sizes the header names for the prover.  The automatic choice now steps down until the table fits the pipeline's
    32-bit offsets: 2^23 BW6-761 terms build at c = 20 and give the variable-base entry point's point (itself checked against the oracle up to
    2^24 terms in tests/test_configs_gpu.py); an explicit c = 21 at that size is still refused."""
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 1 << 23
    pts = syn.device_points("bw6_761_g1", n, 0x5EED4523)
    sc = syn.uniform_scalars("bw6_761_g1", n, 0x5EED4524)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    fb = gpu.FixedBase("bw6_761_g1", d_bases=pts.data_ptr(), n=n, window_bits=0)
    info = fb.info()
    assert info["window_bits"] == 20 and info["n"] == n
    got = co.jac_to_affine(fb.msm_dev(d_sc.data_ptr(), n), "761")
    fb.release()
    assert got == co.jac_to_affine(gpu.msm_dev("bw6_761_g1", pts.data_ptr(), 0, d_sc.data_ptr(), n), "761") and got is not None
    with pytest.raises(RuntimeError):
        gpu.FixedBase("bw6_761_g1", d_bases=pts.data_ptr(), n=n, window_bits=21)
