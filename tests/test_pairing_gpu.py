"""GPU (-m gpu): BLS12-377 pairing product check through the C ABI vs the oracle.

Reference call sites: crates/bls-crypto/src/bls/public.rs:102 (verify, 2 pairs), signature.rs:149 (batch_verify_hashes,
n+1 pairs), batch.rs:83 (strict batches).  The reference only exposes `== Fq12::one()`; here the GT value itself and the
Miller-loop product are additionally compared bit-for-bit with the oracle's arkworks restatement."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the library: torch brings its own HIP runtime; loaded after the library's, it finds no device)
from oracle.py import ecc
from oracle import cpu_oracle as co

pytestmark = pytest.mark.gpu


def _signed_pairs(rng, npairs, bad=None):
    """(n aggregates) -> n+1 pairs of Signature::batch_verify_hashes: (asig, -g2), (H_i, apk_i)...; product == 1."""
    sks = [ecc.random_scalar(rng, ecc.R377) for _ in range(npairs - 1)]
    hs = [ecc.E1_377.mul(ecc.G1_377, rng.next() | 1) for _ in range(npairs - 1)]
    pks = [ecc.E2_377.mul(ecc.G2_377, sk) for sk in sks]
    asig = None
    for sk, h in zip(sks, hs):
        asig = ecc.E1_377.add(asig, ecc.E1_377.mul(h, sk))
    if bad is not None:
        pks[bad] = ecc.E2_377.mul(ecc.G2_377, sks[bad] + 1)
    g1 = [asig] + hs
    g2 = [ecc.E2_377.neg(ecc.G2_377)] + pks
    return g1, g2


def test_gt_values_bit_exact(gpu):
    rng = ecc.SplitMix64(41)
    P = [ecc.E1_377.mul(ecc.G1_377, rng.next()) for _ in range(3)]
    Q = [ecc.E2_377.mul(ecc.G2_377, rng.next()) for _ in range(3)]
    g1, _ = co.pack_g1_377(P)
    g2, _ = co.pack_g2_377(Q)
    # three independent single-pair products: Miller values and GT values
    offs = np.array([0, 1, 2, 3], dtype=np.uint32)
    ml = gpu.pairing_gt(g1, None, g2, None, offs, miller_only=True)
    gt = gpu.pairing_gt(g1, None, g2, None, offs)
    for i in range(3):
        assert np.array_equal(ml[i], co.miller_loop_377(g1[i:i + 1], None, g2[i:i + 1], None))
        assert np.array_equal(gt[i], co.pairing_product_377(g1[i:i + 1], None, g2[i:i + 1], None)[0])
    # one 3-pair product: equals the oracle's shared-squaring multi-Miller loop + one final exponentiation
    gt3 = gpu.pairing_gt(g1, None, g2, None, np.array([0, 3], dtype=np.uint32))
    assert np.array_equal(gt3[0], co.pairing_product_377(g1, None, g2, None)[0])


@pytest.mark.parametrize("k", [1, 2, 3])
def test_single_product_latency_path_bit_exact(gpu, k):
    """ONE product of k <= 3 pairs takes the latency path (csrc/unit_pairing377_wide.hip: operations side by side in three lane
    groups, the Miller loop cut into a point-step wave and three iteration ranges): the Miller value - a product of partial values
    there - and the GT value must be the field elements the oracle's shared-squaring loop and final exponentiation produce
    (what verify / verify_pop compute, crates/bls-crypto/src/bls/public.rs:71-120), also with a pair at infinity."""
    rng = ecc.SplitMix64(4100 + k)
    P = [ecc.E1_377.mul(ecc.G1_377, rng.next()) for _ in range(k)]
    Q = [ecc.E2_377.mul(ecc.G2_377, rng.next()) for _ in range(k)]
    offs = np.array([0, k], dtype=np.uint32)
    for drop in (None, k - 1):
        Pd = [None if j == drop else p for j, p in enumerate(P)]
        g1, i1 = co.pack_g1_377(Pd)
        g2, i2 = co.pack_g2_377(Q)
        ml = gpu.pairing_gt(g1, i1, g2, i2, offs, miller_only=True)
        gt = gpu.pairing_gt(g1, i1, g2, i2, offs)
        assert np.array_equal(ml[0], co.miller_loop_377(g1, i1, g2, i2))
        want, one = co.pairing_product_377(g1, i1, g2, i2)
        assert np.array_equal(gt[0], want)
        assert bool(gpu.pairing_product_is_one(g1, i1, g2, i2)) == bool(one)


def test_bilinearity(gpu):
    a, b = 0x1234567890ABCDEF, 0xFEDCBA0987654321
    P, Q = ecc.G1_377, ecc.G2_377
    g1, _ = co.pack_g1_377([ecc.E1_377.mul(P, a), ecc.E1_377.neg(ecc.E1_377.mul(P, a * b % ecc.R377))])
    g2, _ = co.pack_g2_377([ecc.E2_377.mul(Q, b), Q])
    assert gpu.pairing_product_is_one(g1, None, g2, None)          # e(aP,bQ) * e(-abP,Q) == 1
    g1b, _ = co.pack_g1_377([ecc.E1_377.mul(P, a), ecc.E1_377.neg(ecc.E1_377.mul(P, (a * b + 1) % ecc.R377))])
    assert not gpu.pairing_product_is_one(g1b, None, g2, None)


def test_verify_and_infinity(gpu):
    """crates/bls-crypto/src/bls/public.rs:94-120: e(sig,-g2) * e(H(m),pk) == 1; pairs with an infinite point are skipped."""
    sk = 0x1234567890ABCDEF1234
    Hm = ecc.E1_377.mul(ecc.G1_377, 0xCAFEBABE)
    sig = ecc.E1_377.mul(Hm, sk)
    pk = ecc.E2_377.mul(ecc.G2_377, sk)
    ng2 = ecc.E2_377.neg(ecc.G2_377)
    g1, i1 = co.pack_g1_377([sig, Hm])
    g2, i2 = co.pack_g2_377([ng2, pk])
    assert gpu.pairing_product_is_one(g1, i1, g2, i2)
    assert co.pairing_product_377(g1, i1, g2, i2)[1]
    g2b, _ = co.pack_g2_377([ng2, ecc.E2_377.mul(ecc.G2_377, sk + 1)])
    assert not gpu.pairing_product_is_one(g1, i1, g2b, i2)
    g1c, i1c = co.pack_g1_377([sig, Hm, None])
    g2c, i2c = co.pack_g2_377([ng2, pk, pk])
    assert gpu.pairing_product_is_one(g1c, i1c, g2c, i2c)
    g1d, i1d = co.pack_g1_377([sig, Hm, Hm])
    g2d, i2d = co.pack_g2_377([ng2, pk, None])
    assert gpu.pairing_product_is_one(g1d, i1d, g2d, i2d)
    # empty product is 1
    assert gpu.pairing_product_is_one(np.zeros((0, 12), dtype=np.uint64), None, np.zeros((0, 24), dtype=np.uint64), None)


@pytest.mark.parametrize("npairs", [2, 8, 33])
def test_batch_verify_shape(gpu, npairs):
    """Signature::batch_verify_hashes (signature.rs:125-155): n+1-pair product, accept; corrupt one key, reject; same
    verdicts as the oracle."""
    rng = ecc.SplitMix64(500 + npairs)
    g1p, g2p = _signed_pairs(rng, npairs)
    g1, _ = co.pack_g1_377(g1p)
    g2, _ = co.pack_g2_377(g2p)
    assert gpu.pairing_product_is_one(g1, None, g2, None) and co.pairing_product_377(g1, None, g2, None)[1]
    g1p, g2p = _signed_pairs(ecc.SplitMix64(500 + npairs), npairs, bad=(npairs - 2) // 2)
    g2, _ = co.pack_g2_377(g2p)
    assert not gpu.pairing_product_is_one(g1, None, g2, None)
    assert not co.pairing_product_377(g1, None, g2, None)[1]


def test_many_independent_products(gpu):
    """BASELINE config 3 shape: many independent 2-pair checks in ONE launch, 1-in-8 corrupted; the accept vector must equal
    the oracle's."""
    m = 192
    rng = ecc.SplitMix64(77)
    g1l, g2l, expect = [], [], []
    ng2 = ecc.E2_377.neg(ecc.G2_377)
    for i in range(m):
        sk = ecc.random_scalar(rng, ecc.R377)
        Hm = ecc.E1_377.mul(ecc.G1_377, rng.next() | 1)
        sig = ecc.E1_377.mul(Hm, sk)
        bad = (i % 8) == 3
        pk = ecc.E2_377.mul(ecc.G2_377, sk + (1 if bad else 0))
        g1l += [sig, Hm]
        g2l += [ng2, pk]
        expect.append(0 if bad else 1)
    g1, _ = co.pack_g1_377(g1l)
    g2, _ = co.pack_g2_377(g2l)
    offs = np.arange(0, 2 * m + 1, 2, dtype=np.uint32)
    got = gpu.pairing_product_is_one_batch(g1, None, g2, None, offs)
    assert got.tolist() == expect
    for i in (0, 3, 100):
        assert co.pairing_product_377(g1[2 * i:2 * i + 2], None, g2[2 * i:2 * i + 2], None)[1] == bool(expect[i])
    print("pairing timings:", gpu.pairing_timings())


@pytest.mark.parametrize("exe_name", ["lanes_selftest", "hex_selftest"])
def test_lane_parallel_pieces_match_one_lane_twins(gpu, exe_name):
    """build/lanes_selftest, build/hex_selftest (tools/lanes_selftest.hip for three / six lanes per pairing): every piece of the
    lane-parallel pairing (Fq12 product / square / cyclotomic square / sparse line product / inverse / Frobenius, the point steps,
    truncated and full Miller loops) against its one-lane twin from pairing.h, both on the GPU, canonical Fq12 limbs compared."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "celo-bls-snark-rs_amd", "build", exe_name)
    assert os.path.exists(exe), "run `make -C celo-bls-snark-rs_amd/csrc` (or __graft_entry__.build()) first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and r.stdout.count(" ok") >= 16, r.stdout + r.stderr


def _sampled_gt_check(gpu, g1, i1, g2, i2, offs, got_ok, samples, seed):
    """GT values of `samples` products drawn over the whole call, bit for bit against the oracle, and the oracle's verdicts against the GPU's."""
    m = offs.size - 1
    gt = gpu.pairing_gt(g1, i1, g2, i2, offs)
    pick = np.unique(np.concatenate([np.random.default_rng(seed).integers(0, m, size=samples), [0, 1, m - 1]]))
    for p in pick:
        lo, hi = int(offs[p]), int(offs[p + 1])
        want, one = co.pairing_product_377(g1[lo:hi], None if i1 is None else i1[lo:hi], g2[lo:hi], None if i2 is None else i2[lo:hi])
        assert np.array_equal(gt[p], want), "GT value of product %d" % p
        assert bool(got_ok[p]) == bool(one), "verdict of product %d" % p
    return len(pick)


def test_shared_accumulator_mode_at_scale_distinct_products(gpu):
    """>= 16384 products of <= 4 pairs in one call take the one-group-per-product path (pairing_lanes.h miller_multi: one accumulator per
    product).  VERDICT r5 item 4: rounds 1-5 tiled EIGHT products 2500 times - a data-dependent fault cannot show there.  Here 20480 DISTINCT
    products, all generated on the device (synthetic.verify_products / random_pair_products): 16384 verify shapes with their own (sig, H, pk)
    each (every 97th carries a foreign signature), 2048 three-pair products of unrelated points (never 1), and 2048 verify shapes with a THIRD pair
    that contributes 1 because one of its points is at infinity (skipped, as ark-ec's miller_loop does).  Accept vector against construction;
    GT values of 67 sampled products bit-exact against the oracle."""
    from celo_bls_snark_rs_amd import synthetic as syn
    a1, a2, _, ea = syn.verify_products(16384, 0x60D0001)
    b1, b2, _ = syn.random_pair_products(2048, 3, 0x60D0002)
    c1v, c2v, _, ec = syn.verify_products(2048, 0x60D0003)
    x1, x2, _ = syn.random_pair_products(2048, 1, 0x60D0004)
    c1 = np.empty((2048 * 3, 12), dtype=np.uint64); c2 = np.empty((2048 * 3, 24), dtype=np.uint64)
    c1[0::3] = c1v[0::2]; c1[1::3] = c1v[1::2]; c1[2::3] = x1
    c2[0::3] = c2v[0::2]; c2[1::3] = c2v[1::2]; c2[2::3] = x2
    g1 = np.concatenate([a1, b1, c1]); g2 = np.concatenate([a2, b2, c2])
    offs = np.concatenate([np.arange(0, 2 * 16384, 2), 2 * 16384 + np.arange(0, 3 * 2048, 3), 2 * 16384 + 3 * 2048 + np.arange(0, 3 * 2048 + 1, 3)]).astype(np.uint32)
    i1 = np.zeros(g1.shape[0], dtype=np.uint8); i2 = np.zeros(g2.shape[0], dtype=np.uint8)
    third = 2 * 16384 + 3 * 2048 + 2 + 3 * np.arange(2048)
    i1[third[0::2]] = 1                                               # G1 at infinity in the third pair of every other such product ...
    i2[third[1::2]] = 1                                               # ... G2 at infinity in the others
    expect = ea + [0] * 2048 + ec
    assert sum(expect) > 18000 and expect.count(0) > 2048 + 150
    got = gpu.pairing_product_is_one_batch(g1, i1, g2, i2, offs)
    assert got.astype(int).tolist() == expect
    assert _sampled_gt_check(gpu, g1, i1, g2, i2, offs, got, 64, 11) >= 60


def test_two_pair_products_of_unrelated_points_at_scale(gpu):
    """20480 DISTINCT two-pair products with no shared G2 point (k_miller_product_slots<LPH377, 2>, both pairs walk their own point): none is 1,
    GT values of 67 sampled products bit-exact against the oracle; and the same points through the one-group-per-PAIR path (6000 products:
    k_miller_slots + k_gt_product_lanes) give the same GT values."""
    from celo_bls_snark_rs_amd import synthetic as syn
    g1, g2, offs = syn.random_pair_products(20480, 2, 0x60D0010)
    got = gpu.pairing_product_is_one_batch(g1, None, g2, None, offs)
    assert not got.any()
    assert _sampled_gt_check(gpu, g1, None, g2, None, offs, got, 64, 12) >= 60
    small = gpu.pairing_gt(g1[:12000], None, g2[:12000], None, offs[:6001])
    big = gpu.pairing_gt(g1, None, g2, None, offs)
    assert np.array_equal(small, big[:6000])


@pytest.mark.wall_clock(600)
def test_verify_shaped_products_81920_distinct(gpu):
    """The bench's pairing leg, product for product distinct (prepared lines for the shared -g2 + k_miller_prepared_slots + k_final_exp_slots):
    81920 (sig_b, H_b, pk_b) triples of their own, every 97th with a foreign signature: accept vector against construction, GT values of 67
    sampled products bit-exact against the oracle."""
    from celo_bls_snark_rs_amd import synthetic as syn
    g1, g2, offs, expect = syn.verify_products(81920, 0x60D0020)
    got = gpu.pairing_product_is_one_batch(g1, None, g2, None, offs)
    assert got.astype(int).tolist() == expect and expect.count(0) == len(range(1, 81919, 97))
    assert _sampled_gt_check(gpu, g1, None, g2, None, offs, got, 64, 13) >= 60


def test_bilinearity_on_reference_held_points(gpu, golden):
    """A check that shares nothing with the Miller-loop / final-exponentiation restatements: for points the REFERENCE holds (a G1
    hash-to-curve vector, crates/bls-crypto/src/hash_to_curve/mod.rs:412-455, and a validator key of its Groth16 FFI test,
    crates/bls-snark-sys/src/snark/mod.rs:56) and known scalars a, b: e(aP, bQ) * e(-abP, Q) == 1 and e(aP, bQ) * e(-(ab+1)P, Q) != 1,
    on the GPU and on the oracle.  Scalar multiplication is pinned separately (cofactor clearing of the hash-to-curve vectors)."""
    P = ecc.deser_point(ecc.E1_377, bytes.fromhex(golden["hash_to_curve"]["g1_compat"]["points"][3]))
    Q = ecc.deser_point(ecc.E2_377, bytes.fromhex(golden["groth16_bw6_761"]["first_pubkeys"])[:96])
    assert ecc.E1_377.in_subgroup(P) and ecc.E2_377.in_subgroup(Q)
    rng = ecc.SplitMix64(777)
    for _ in range(3):
        a, b = ecc.random_scalar(rng, ecc.R377), ecc.random_scalar(rng, ecc.R377)
        aP, bQ = ecc.E1_377.mul(P, a), ecc.E2_377.mul(Q, b)
        for delta, want in ((0, True), (1, False)):
            m = ecc.E1_377.neg(ecc.E1_377.mul(P, (a * b + delta) % ecc.R377))
            g1, i1 = co.pack_g1_377([aP, m])
            g2, i2 = co.pack_g2_377([bQ, Q])
            assert gpu.pairing_product_is_one(g1, i1, g2, i2) == want
            assert bool(co.pairing_product_377(g1, i1, g2, i2)[1]) == want


def test_prepared_first_pair_path_at_scale(gpu):
    """>= 16384 two-pair products whose first pair shares one G2 point (the verify / Batch::verify shape e(S, -g2) * e(H, P)) take the
    prepared-lines kernel (the shared point's 69 line triples computed once).  20480 products tiled from 8 signed messages (two with a
    foreign key), plus: a product whose S is the identity (pair skipped), a one-pair product, an empty product.  Verdicts equal the
    oracle's on the distinct products; the same call with ONE product's first point changed falls back to the generic kernel and
    returns the same verdicts for all the others."""
    rng = ecc.SplitMix64(2468)
    ng2 = ecc.E2_377.neg(ecc.G2_377)
    trip = []
    for i in range(8):
        sk = ecc.random_scalar(rng, ecc.R377)
        Hm = ecc.E1_377.mul(ecc.G1_377, rng.next() | 1)
        good = i not in (2, 5)
        trip.append((ecc.E1_377.mul(Hm, sk), Hm, ecc.E2_377.mul(ecc.G2_377, sk if good else sk + 7), good))
    m = 20480
    g1l, g2l, offs, expect = [], [], [0], []
    for i in range(8):
        S, Hm, P, good = trip[i]
        g1l += [S, Hm]; g2l += [ng2, P]
    g1b, i1b = co.pack_g1_377(g1l); g2b, i2b = co.pack_g2_377(g2l)
    g1 = np.tile(g1b, (m // 8, 1)); g2 = np.tile(g2b, (m // 8, 1))
    i1 = np.zeros(2 * m, dtype=np.uint8); i2 = np.zeros(2 * m, dtype=np.uint8)
    expect = [int(trip[i % 8][3]) for i in range(m)]
    offs = np.arange(0, 2 * m + 1, 2, dtype=np.uint32)
    # product 11: S = identity -> only e(H, P) remains: not 1.  Product 12: both pairs skipped -> 1.
    i1[22] = 1; expect[11] = 0
    i1[24] = 1; i1[25] = 1; expect[12] = 1
    got = gpu.pairing_product_is_one_batch(g1, i1, g2, i2, offs)
    assert got.tolist() == expect
    for p in list(range(8)) + [11, 12]:
        lo = 2 * p
        assert bool(co.pairing_product_377(g1[lo:lo + 2], i1[lo:lo + 2], g2[lo:lo + 2], i2[lo:lo + 2])[1]) == bool(expect[p])
    # ragged: a one-pair product and an empty product in the middle (offsets no longer uniform)
    offs2 = offs.copy().astype(np.int64)
    offs2[101:] -= 1                         # product 100 keeps only its first pair e(S, -g2): not 1
    offs2[201:] -= 2                         # product 200 is empty: 1
    keep = np.ones(2 * m, dtype=bool); keep[201] = False; keep[400] = False; keep[401] = False
    got2 = gpu.pairing_product_is_one_batch(g1[keep], i1[keep], g2[keep], i2[keep], offs2.astype(np.uint32))
    exp2 = list(expect); exp2[100] = 0; exp2[200] = 1
    assert got2.tolist() == exp2
    # one product with another first point: the generic shared-accumulator kernel runs instead; every verdict is unchanged
    g2c = g2.copy()
    g2c[2 * 300] = g2c[2 * 300 + 1]
    got3 = gpu.pairing_product_is_one_batch(g1, i1, g2c, i2, offs)
    exp3 = list(expect); exp3[300] = int(co.pairing_product_377(g1[600:602], None, g2c[600:602], None)[1])
    assert got3.tolist() == exp3


def test_small_batches_on_the_latency_path(gpu):
    """Up to 768 products of <= 3 pairs take the latency path a block each (what concurrent verify callers, combined into one
    launch, or a small batch_verify bring): mixed pair counts, an empty product, a corrupted one; verdicts as constructed and GT
    values against the oracle on a sample."""
    rng = ecc.SplitMix64(977)
    g1l, g2l, offs, want = [], [], [0], []
    for p in range(40):
        npairs = (p % 3) + 2 if p != 7 else 0                       # 2, 3 (rejects: a 4-pair product would leave the path), one empty
        if npairs == 0:
            offs.append(offs[-1]); want.append(True); continue
        if npairs == 4:
            npairs = 3
        a, b = _signed_pairs(rng, min(npairs, 3), bad=(0 if p % 5 == 4 else None))
        g1l += a; g2l += b
        offs.append(offs[-1] + len(a)); want.append(p % 5 != 4)
    g1, i1 = co.pack_g1_377(g1l)
    g2, i2 = co.pack_g2_377(g2l)
    offs = np.array(offs, dtype=np.uint32)
    got = gpu.pairing_product_is_one_batch(g1, i1, g2, i2, offs)
    assert [bool(x) for x in got.tolist()] == want
    gt = gpu.pairing_gt(g1, i1, g2, i2, offs)
    for p in (0, 4, 8, 39):
        lo, hi = int(offs[p]), int(offs[p + 1])
        assert np.array_equal(gt[p], co.pairing_product_377(g1[lo:hi], i1[lo:hi], g2[lo:hi], i2[lo:hi])[0])


def test_medium_batches_final_exponentiation_side_by_side(gpu):
    """769 ... 3072 products leave the latency path and take the final exponentiation with three products per wave
    (csrc/unit_pairing377_wide.hip k377_w3_final_products, round 4): verdicts as constructed and GT values bit for bit against the
    oracle on a sample - at a product count that is not a multiple of three (a surplus super-group) and with an empty product."""
    base = 40
    rng = ecc.SplitMix64(3072)
    pairs = [_signed_pairs(rng, 2, bad=(0 if p % 7 == 3 else None)) for p in range(base)]
    m = 1001
    g1l, g2l, offs, want = [], [], [0], []
    for p in range(m):
        if p == 500:
            offs.append(offs[-1]); want.append(True); continue          # empty product: 1
        a, b = pairs[p % base]
        g1l += a; g2l += b
        offs.append(offs[-1] + 2); want.append((p % base) % 7 != 3)
    g1, i1 = co.pack_g1_377(g1l)
    g2, i2 = co.pack_g2_377(g2l)
    offs = np.array(offs, dtype=np.uint32)
    got = gpu.pairing_product_is_one_batch(g1, i1, g2, i2, offs)
    assert [bool(x) for x in got.tolist()] == want
    gt = gpu.pairing_gt(g1, i1, g2, i2, offs)
    for p in (0, 3, 499, 501, 999, 1000):
        lo, hi = int(offs[p]), int(offs[p + 1])
        assert np.array_equal(gt[p], co.pairing_product_377(g1[lo:hi], i1[lo:hi], g2[lo:hi], i2[lo:hi])[0]), p
    one = co.pairing_product_377(g1[:0], None, g2[:0], None)[0]
    assert np.array_equal(gt[500], one)


@pytest.mark.parametrize("m", [1003, 3101])
def test_medium_batches_split_by_iteration_range(gpu, m):
    """769 ... 5120 verify-shaped products of exactly two pairs take k_miller_prepared_split_slots (late round 4): every product cut in
    two by iteration range, F_63 = F_h^(2^(63 - h)) * G.  Verdicts as constructed, Miller AND GT values bit for bit against the oracle on
    a sample, with a pair flagged as the identity in some products (that pair's line is 1) and a product count that is not a multiple
    of the ten groups of a wave.  m = 1003: final exponentiation with three groups per product; m = 3101: with two (k377_w2_final_products)."""
    base = 30
    rng = ecc.SplitMix64(5120)
    pairs = [_signed_pairs(rng, 2, bad=(0 if p % 6 == 2 else None)) for p in range(base)]
    g1l, g2l, want = [], [], []
    for p in range(m):
        a, b = pairs[p % base]
        g1l += a; g2l += b
        want.append((p % base) % 6 != 2)
    g1, i1 = co.pack_g1_377(g1l)
    g2, i2 = co.pack_g2_377(g2l)
    offs = np.arange(0, 2 * m + 1, 2, dtype=np.uint32)
    got = gpu.pairing_product_is_one_batch(g1, i1, g2, i2, offs)
    assert [bool(x) for x in got.tolist()] == want
    # identity flags: product 7's second pair and product 500's first pair are left out (each then is a one-pair product)
    i1b, i2b = i1.copy(), i2.copy()
    i2b[2 * 7 + 1] = 1
    i1b[2 * 500] = 1
    gt = gpu.pairing_gt(g1, i1b, g2, i2b, offs)
    ml = gpu.pairing_gt(g1, i1b, g2, i2b, offs, miller_only=True)
    for p in (0, 2, 7, 499, 500, m - 1):
        lo, hi = 2 * p, 2 * p + 2
        assert np.array_equal(ml[p], co.miller_loop_377(g1[lo:hi], i1b[lo:hi], g2[lo:hi], i2b[lo:hi])), p
        assert np.array_equal(gt[p], co.pairing_product_377(g1[lo:hi], i1b[lo:hi], g2[lo:hi], i2b[lo:hi])[0]), p
