"""GPU (-m gpu): every BASELINE.json configuration at its stated size, plus the library's threading / multi-device contract.

cfg1  64-validator aggregate signature verify through the bls-snark-sys ABI (crates/bls-crypto/examples/simple_signature.rs:31-64)
cfg3  4096 batches x 256 signers, 1 % corrupted, through the device-chained Batch::verify (crates/bls-crypto/src/bls/batch.rs:44-84);
      accept vector as constructed + the oracle on a 64-batch sample
cfg4  BW6-761 G1 MSM at the per-GPU shard size 2^21 of the 2^24 / 8-GPU job (crates/epoch-snark/src/api/prover.rs:78) vs the oracle
cfg5  G1 MSM 2^22 + G2 MSM 2^22 + 2^14 Miller loops issued concurrently; G2 2^22 vs the oracle; concurrent == sequential
Parity is bit-exact on affine-normalised group elements / accept bits."""
import ctypes as C
import threading
import time
import numpy as np
import pytest
import torch
from oracle.py import ecc
from oracle import cpu_oracle as co
from tests import helpers as H
from tests.test_seam_a import sys_lib, _deser, _ser  # noqa: F401  (fixture + helpers of the Seam A tests)

pytestmark = pytest.mark.gpu


def _threads():
    return max(1, min(64, co.lib().orc_hardware_threads()))


# ------------------------------------------------------------------------------------------------ cfg1
@pytest.mark.parametrize("composite,cip22", [(False, False), (True, True)])
def test_cfg1_64_validator_aggregate_signature(sys_lib, gpu, composite, cip22):
    """simple_signature.rs with 64 keys: every validator signs the message, signatures and keys are aggregated, ONE
    verify_signature accepts; a wrong message, a missing signer and a foreign key are rejected."""
    lib = sys_lib
    for f in ("sign_message", "verify_signature"):
        getattr(lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(composite), C.c_bool(cip22)
    rng = ecc.SplitMix64(64)
    sks, pks, sigs = [], [], []
    msg, extra = b"hello", b""
    for _ in range(64):
        sk = _deser(lib, "deserialize_private_key", ecc.random_scalar(rng, ecc.R377).to_bytes(32, "little"))
        pk = C.c_void_p()
        assert lib.private_key_to_public_key(sk, C.byref(pk))
        s = C.c_void_p()
        assert lib.sign_message(sk, msg, C.c_int(len(msg)), extra, C.c_int(0), CF, C22, C.byref(s))
        sks.append(sk); pks.append(pk); sigs.append(s)
    apk, asig = C.c_void_p(), C.c_void_p()
    assert lib.aggregate_public_keys((C.c_void_p * 64)(*[p.value for p in pks]), C.c_int(64), C.byref(apk))
    assert lib.aggregate_signatures((C.c_void_p * 64)(*[s.value for s in sigs]), C.c_int(64), C.byref(asig))
    ok = C.c_bool(False)
    assert lib.verify_signature(apk, msg, C.c_int(5), extra, C.c_int(0), asig, CF, C22, C.byref(ok)) and ok.value
    assert lib.verify_signature(apk, b"hellO", C.c_int(5), extra, C.c_int(0), asig, CF, C22, C.byref(ok)) and not ok.value
    asig63 = C.c_void_p()
    assert lib.aggregate_signatures((C.c_void_p * 63)(*[s.value for s in sigs[:63]]), C.c_int(63), C.byref(asig63))
    assert lib.verify_signature(apk, msg, C.c_int(5), extra, C.c_int(0), asig63, CF, C22, C.byref(ok)) and not ok.value
    assert lib.verify_signature(pks[0], msg, C.c_int(5), extra, C.c_int(0), asig, CF, C22, C.byref(ok)) and not ok.value
    # the aggregate key survives the wire format (what the example prints and what a verifier would receive)
    apk2 = _deser(lib, "deserialize_public_key", _ser(lib, "serialize_public_key", apk))
    assert lib.verify_signature(apk2, msg, C.c_int(5), extra, C.c_int(0), asig, CF, C22, C.byref(ok)) and ok.value


# ------------------------------------------------------------------------------------------------ cfg3
def _oracle_batch_verdict(pk, sig, ex, h, ng2):
    """Batch::verify of one batch on the oracle: P = sum e_j pk_j, S = sum e_j sig_j, e(S, -g2) * e(H, P) == 1."""
    P = co.jac_to_affine(co.msm("bls12_377_g2", pk, None, ex, threads=4), "g2_377")
    S = co.jac_to_affine(co.msm("bls12_377_g1", sig, None, ex, threads=4), "g1_377")
    g1, i1 = co.pack_g1_377([S, None])
    g1[1] = h
    g2, i2 = co.pack_g2_377([None, P])
    g2[0] = ng2
    i1[1] = 0; i2[0] = 0
    return co.pairing_product_377(g1, i1, g2, i2)[1]


def test_cfg3_4096_batches_of_256_with_one_percent_corrupted(gpu):
    from celo_bls_snark_rs_amd import synthetic as syn
    m, n = 4096, 256
    rng = np.random.default_rng(33)
    corrupt = np.sort(rng.choice(m, size=41, replace=False))
    corrupt[0] = 3                                           # make sure the oracle sample below holds rejected batches
    corrupt[1] = 40
    corrupt = np.unique(corrupt)
    w = syn.valid_batches(m, n, 0x5EED0300, corrupt)
    ex = syn.batch_exponents(m * n, 0x5EED0301)
    d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
    ng2 = syn.neg_g2_limbs()
    got = gpu.batch_verify_dev(w["pk"].data_ptr(), w["sig"].data_ptr(), d_ex.data_ptr(), w["offsets"], w["hash"].data_ptr(), ng2)
    assert got.tolist() == w["expect"].tolist(), "accept vector differs from the constructed one"
    assert int(got.sum()) == m - len(corrupt)
    # oracle on the first 64 batches (two of them corrupted)
    pk = w["pk"].view(m * n, 24)[: 64 * n].cpu().numpy().view(np.uint64)
    sg = w["sig"].view(m * n, 12)[: 64 * n].cpu().numpy().view(np.uint64)
    hh = w["hash"].view(m, 12)[:64].cpu().numpy().view(np.uint64)
    want = [_oracle_batch_verdict(pk[b * n:(b + 1) * n], sg[b * n:(b + 1) * n], ex[b * n:(b + 1) * n], hh[b], ng2) for b in range(64)]
    assert [bool(x) for x in got[:64]] == want
    assert want.count(False) == 2
    # the host-buffer form of the same entry point on those 64 batches
    got_h = gpu.batch_verify(pk, sg, ex[: 64 * n], w["offsets"][:65], hh, ng2)
    assert got_h.tolist() == got[:64].tolist()


def test_batch_verify_ragged_and_degenerate(gpu):
    """ragged batch sizes (1 .. 300 signers), an all-zero exponent batch and an empty batch: same verdicts as the oracle."""
    from celo_bls_snark_rs_amd import synthetic as syn
    sizes = [1, 7, 300, 64, 0, 33]
    nmax = 300
    m = len(sizes)
    w = syn.valid_batches(m, nmax, 77, [1])
    ex_full = syn.batch_exponents(m * nmax, 78)
    pk_f = w["pk"].view(m * nmax, 24).cpu().numpy().view(np.uint64)
    sg_f = w["sig"].view(m * nmax, 12).cpu().numpy().view(np.uint64)
    hh = w["hash"].view(m, 12).cpu().numpy().view(np.uint64)
    pk = np.concatenate([pk_f[b * nmax: b * nmax + k] for b, k in enumerate(sizes)])
    sg = np.concatenate([sg_f[b * nmax: b * nmax + k] for b, k in enumerate(sizes)])
    ex = np.concatenate([ex_full[b * nmax: b * nmax + k] for b, k in enumerate(sizes)])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    ex[offs[3]: offs[4]] = 0                                  # batch 3: every exponent zero -> both sums are the identity -> accepts
    ng2 = syn.neg_g2_limbs()
    got = gpu.batch_verify(pk, sg, ex, offs, hh, ng2)
    want = []
    for b, k in enumerate(sizes):
        lo, hi = int(offs[b]), int(offs[b + 1])
        want.append(True if k == 0 else _oracle_batch_verdict(pk[lo:hi], sg[lo:hi], ex[lo:hi], hh[b], ng2))
    assert [bool(x) for x in got] == want
    assert want == [True, False, True, True, True, True]


def test_batch_verify_oversized_batches_on_the_chained_path(gpu):
    """batch_verify_bls12_377[_dev] with a batch of more than 1024 signers (the batched MSM path's per-instance limit): the chained
    call takes the big pipeline per instance and still returns the oracle's verdicts (it returned rc 2)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    m, n = 3, 1300
    w = syn.valid_batches(m, n, 0x5EED0311, [2])
    ex = syn.batch_exponents(m * n, 0x5EED0312)
    d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
    ng2 = syn.neg_g2_limbs()
    got = gpu.batch_verify_dev(w["pk"].data_ptr(), w["sig"].data_ptr(), d_ex.data_ptr(), w["offsets"], w["hash"].data_ptr(), ng2)
    assert got.tolist() == [1, 1, 0] == w["expect"].tolist()
    pk = w["pk"].view(m * n, 24).cpu().numpy().view(np.uint64)
    sg = w["sig"].view(m * n, 12).cpu().numpy().view(np.uint64)
    hh = w["hash"].view(m, 12).cpu().numpy().view(np.uint64)
    want = [_oracle_batch_verdict(pk[b * n:(b + 1) * n], sg[b * n:(b + 1) * n], ex[b * n:(b + 1) * n], hh[b], ng2) for b in range(m)]
    assert [bool(x) for x in got] == want
    # host-buffer form; ragged with an empty batch between two oversized ones, and a call whose batches are all empty
    sizes = [1025, 0, 1300]
    offs = np.array([0, 1025, 1025, 2325], dtype=np.uint32)
    pk2 = np.concatenate([pk[:1025], pk[2 * n: 3 * n]]); sg2 = np.concatenate([sg[:1025], sg[2 * n: 3 * n]]); ex2 = np.concatenate([ex[:1025], ex[2 * n: 3 * n]])
    got2 = gpu.batch_verify(pk2, sg2, ex2, offs, hh, ng2)
    want2 = [_oracle_batch_verdict(pk2[:1025], sg2[:1025], ex2[:1025], hh[0], ng2), True, False]
    assert [bool(x) for x in got2] == want2 and want2[0] is True
    got3 = gpu.batch_verify(pk2[:0], sg2[:0], ex2[:0], np.zeros(3, dtype=np.uint32), hh[:2], ng2)
    assert [bool(x) for x in got3] == [True, True]


# ------------------------------------------------------------------------------------------------ cfg4
def test_cfg4_bw6_761_g1_shard_size_vs_oracle(gpu):
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 1 << 21
    bases = syn.device_points("bw6_761_g1", n, 0x5EED0400)
    sc = syn.uniform_scalars("bw6_761_g1", n, 0x5EED0401)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    got = co.jac_to_affine(gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n), "761")
    h = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    exp = co.jac_to_affine(co.msm("bw6_761_g1", h, None, sc, threads=_threads()), "761")
    assert got == exp and got is not None
    # witness-like scalars (about 60 % zeros and ones: arkworks skips zeros and adds ones without a bucket) at 2^18
    k = 1 << 18
    sw = syn.witness_like_scalars("bw6_761_g1", k, 0x5EED0402)
    d_sw = torch.from_numpy(sw.view(np.int64)).cuda()
    got = co.jac_to_affine(gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sw.data_ptr(), k), "761")
    assert got == co.jac_to_affine(co.msm("bw6_761_g1", h[:k], None, sw, threads=_threads()), "761")


def test_cfg4_whole_job_2p24_on_one_gpu_vs_oracle_over_the_8_index_shards(gpu):
    """BASELINE config 4 AS NAMED - the epoch-snark prover's BW6-761 G1 MSM of 2^24 terms (crates/epoch-snark/src/api/prover.rs:78) - run whole on
    ONE GPU (3.2 GB of bases + 0.8 GB of scalars fit; it is also the only stand-in for the 8-GPU fold while no node exists): the device result
    against the oracle run over the SAME 8 index shards the 8-GPU job would cut (one oracle call per shard, side by side on the host's cores, the
    eight partial points added with the big-integer group law), for uniform and for witness-like scalars; and the 8-engine in-process fold of the
    same shards (msm_bw6_761_g1_multi_dev on device 0) against both."""
    import threading
    from celo_bls_snark_rs_amd import synthetic as syn
    n, shards = 1 << 24, 8
    per = n // shards
    bases = syn.device_points("bw6_761_g1", n, 0x5EED0424)
    h = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    T = max(1, min(24, co.lib().orc_hardware_threads() // shards))
    for kind, sc in (("uniform", syn.uniform_scalars("bw6_761_g1", n, 0x5EED0425)), ("witness-like", syn.witness_like_scalars("bw6_761_g1", n, 0x5EED0426))):
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        got = co.jac_to_affine(gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n), "761")
        parts = [None] * shards

        def run(i):
            parts[i] = co.jac_to_affine(co.msm("bw6_761_g1", h[i * per:(i + 1) * per], None, sc[i * per:(i + 1) * per], threads=T), "761")
        th = [threading.Thread(target=run, args=(i,)) for i in range(shards)]
        for t in th: t.start()
        for t in th: t.join()
        exp = None
        for P in parts:
            exp = ecc.E1_761.add(exp, P)
        assert got == exp and got is not None, kind
        if kind == "uniform":
            bp = [bases.data_ptr() + i * per * 24 * 8 for i in range(shards)]
            sp = [d_sc.data_ptr() + i * per * 6 * 8 for i in range(shards)]
            multi = gpu.msm_multi_dev("bw6_761_g1", [0] * shards, bp, None, sp, [per] * shards)
            assert co.jac_to_affine(multi, "761") == exp
        del d_sc


# ------------------------------------------------------------------------------------------------ cfg5
def test_cfg5_mixed_g1_g2_msm_and_miller_loops_concurrently(gpu):
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 1 << 22
    b1 = syn.device_points("bls12_377_g1", n, 0x5EED0500)
    b2 = syn.device_points("bls12_377_g2", n, 0x5EED0501)
    s1 = syn.uniform_scalars("bls12_377_g1", n, 0x5EED0502)
    s2 = syn.uniform_scalars("bls12_377_g2", n, 0x5EED0503)
    d1 = torch.from_numpy(s1.view(np.int64)).cuda()
    d2 = torch.from_numpy(s2.view(np.int64)).cuda()
    # 2^14 independent Miller loops as 8192 DISTINCT two-pair products (their own sig, H, pk each; every 97th carries a foreign signature)
    mprod = 8192
    g1, g2, offs, expect = syn.verify_products(mprod, 0x5EED0504)

    def leg_g1():
        return gpu.msm_dev("bls12_377_g1", b1.data_ptr(), 0, d1.data_ptr(), n)

    def leg_g2():
        return gpu.msm_dev("bls12_377_g2", b2.data_ptr(), 0, d2.data_ptr(), n)

    def leg_pairing():
        return gpu.pairing_product_is_one_batch(g1, None, g2, None, offs)

    legs = [leg_g1, leg_g2, leg_pairing]
    for f in legs:
        f()                                                  # warm-up (arenas)
    t0 = time.perf_counter()
    seq = [f() for f in legs]
    t_seq = time.perf_counter() - t0
    res = [None] * 3

    def run(i):
        res[i] = legs[i]()
    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_con = time.perf_counter() - t0
    print("cfg5 legs sequential %.1f ms, concurrent %.1f ms" % (t_seq * 1e3, t_con * 1e3))
    assert co.jac_to_affine(res[0], "g1_377") == co.jac_to_affine(seq[0], "g1_377")
    assert co.jac_to_affine(res[1], "g2_377") == co.jac_to_affine(seq[1], "g2_377")
    assert res[2].tolist() == seq[2].tolist() == expect
    for p in (0, 1, 2, 97, 98, 4000, 8190, 8191):            # ... and the oracle's verdicts on products drawn over the call (1 and 98 are foreign signatures)
        assert bool(co.pairing_product_377(g1[2 * p:2 * p + 2], None, g2[2 * p:2 * p + 2], None)[1]) == bool(expect[p])
    assert t_con < 1.15 * t_seq                              # separate engines and streams: not slower than back to back (measured: 0.87-0.9x)
    # G2 at 2^22 against the oracle (G1 at 2^22: tests/test_msm_gpu.py)
    h2 = b2.cpu().numpy().view(np.uint64).reshape(n, 24)
    assert co.jac_to_affine(seq[1], "g2_377") == co.jac_to_affine(co.msm("bls12_377_g2", h2, None, s2, threads=_threads()), "g2_377")


# ------------------------------------------------------------------------------------------------ threading / devices
def test_multi_device_entry_points_shard_and_fold(gpu):
    """msm_*_multi on a 1-GPU box: the device listed twice / three times = that many engines on it, same result as one call."""
    from celo_bls_snark_rs_amd import synthetic as syn
    assert gpu.device_count() >= 1
    for n in (5, 1000, 1 << 17):
        pts = syn.device_points("bls12_377_g1", n, 900 + n)
        sc = syn.uniform_scalars("bls12_377_g1", n, 901 + n)
        h = pts.cpu().numpy().view(np.uint64).reshape(n, 12)
        exp = co.jac_to_affine(co.msm("bls12_377_g1", h, None, sc, threads=8), "g1_377")
        assert co.jac_to_affine(gpu.msm_multi("bls12_377_g1", [0, 0, 0], h, None, sc), "g1_377") == exp
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        cut = n // 3
        got = gpu.msm_multi_dev("bls12_377_g1", [0, 0], [pts.data_ptr(), pts.data_ptr() + cut * 96], None,
                                [d_sc.data_ptr(), d_sc.data_ptr() + cut * 32], [cut, n - cut])
        assert co.jac_to_affine(got, "g1_377") == exp
    n = 3000
    for group, kind in (("bls12_377_g2", "g2_377"), ("bw6_761_g1", "761")):
        pts = syn.device_points(group, n, 950)
        sc = syn.uniform_scalars(group, n, 951)
        h = pts.cpu().numpy().view(np.uint64).reshape(n, 24)
        exp = co.jac_to_affine(co.msm(group, h, None, sc, threads=8), kind)
        assert co.jac_to_affine(gpu.msm_multi(group, [0, 0], h, None, sc), kind) == exp
    with pytest.raises(RuntimeError):
        gpu.msm_multi("bls12_377_g1", [0, 99], h[:, :12], None, sc[:, :4])


def test_entry_points_from_secondary_threads(gpu):
    """HIP's current device is per thread: calls from threads that never called init, and from a thread bound with
    celo_amd_use_device, must land on the selected device; an out-of-range device is refused."""
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, 200, 31)
    sc = H.seeded_scalars(200, 32, ecc.R377)
    xy, inf = co.pack_g1_377(pts)
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=2), "g1_377")
    out = {}

    def plain():
        out["plain"] = co.jac_to_affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377")

    def bound():
        gpu.use_device(0)
        out["bound"] = co.jac_to_affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377")
        try:
            gpu.use_device(gpu.device_count())
            out["oob"] = "accepted"
        except RuntimeError:
            out["oob"] = "refused"
        out["after"] = co.jac_to_affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377")     # the failed bind changed nothing
    for f in (plain, bound):
        t = threading.Thread(target=f)
        t.start(); t.join()
    assert out == {"plain": exp, "bound": exp, "oob": "refused", "after": exp}


def test_concurrent_mixed_entry_points(gpu):
    """no global lock: 12 host threads issue MSMs (three groups), batched MSMs, pairing batches and NTTs at once; every result
    equals the one computed alone."""
    from oracle.py import ntt as ontt
    rng = ecc.SplitMix64(5150)
    p1 = H.seeded_points(ecc.E1_377, ecc.G1_377, 300, 1); x1, i1 = co.pack_g1_377(p1)
    p2 = H.seeded_points(ecc.E2_377, ecc.G2_377, 120, 2); x2, i2 = co.pack_g2_377(p2)
    s1 = H.scalars_np(H.seeded_scalars(300, 3, ecc.R377), 4)
    s2 = H.scalars_np(H.seeded_scalars(120, 4, ecc.R377), 4)
    offs = np.array([0, 100, 300], dtype=np.uint32)
    sk = ecc.random_scalar(rng, ecc.R377)
    Hm = ecc.E1_377.mul(ecc.G1_377, 0xBEEF)
    g1, _ = co.pack_g1_377([ecc.E1_377.mul(Hm, sk), Hm] * 40)
    g2, _ = co.pack_g2_377([ecc.E2_377.neg(ecc.G2_377), ecc.E2_377.mul(ecc.G2_377, sk)] * 40)
    po = np.arange(0, 81, 2, dtype=np.uint32)
    w = ontt.root_of_unity(10)
    xs = co.to_mont([ecc.random_scalar(rng, ecc.Q377) for _ in range(1024)], ecc.Q377)
    wm = co.to_mont([w], ecc.Q377)[0]
    jobs = {
        "g1": lambda: gpu.msm("bls12_377_g1", x1, i1, s1).tolist(),
        "g2": lambda: gpu.msm("bls12_377_g2", x2, i2, s2).tolist(),
        "batch": lambda: gpu.msm_batch("bls12_377_g1", x1, i1, s1, offs).tolist(),
        "pair": lambda: gpu.pairing_product_is_one_batch(g1, None, g2, None, po).tolist(),
        "ntt": lambda: gpu.ntt(xs, 10, wm).tolist(),
    }
    alone = {k: f() for k, f in jobs.items()}
    assert alone["pair"] == [1] * 40
    assert np.array_equal(np.array(alone["ntt"], dtype=np.uint64), co.ntt_fq377(xs, 10, w))
    results, errs = [], []

    def worker(i):
        try:
            for r in range(3):
                k = list(jobs)[(i + r) % len(jobs)]
                results.append((k, jobs[k]()))
        except Exception as e:                               # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert len(results) == 36
    for k, r in results:
        if k in ("g1", "g2"):                                # Jacobian representatives may differ between engines' schedules: compare the point
            kind = "g1_377" if k == "g1" else "g2_377"
            assert co.jac_to_affine(np.array(r, dtype=np.uint64), kind) == co.jac_to_affine(np.array(alone[k], dtype=np.uint64), kind)
        elif k == "batch":
            for a, b in zip(r, alone[k]):
                assert co.jac_to_affine(np.array(a, dtype=np.uint64), "g1_377") == co.jac_to_affine(np.array(b, dtype=np.uint64), "g1_377")
        else:
            assert r == alone[k], k
