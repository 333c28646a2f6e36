"""GPU (-m gpu): MSM parity — the HIP pipeline through the C ABI vs the oracle on the same seeded inputs.

Reference call sites: crates/bls-crypto/src/bls/signature.rs:85 (G1), public.rs:61 (G2),
ark_groth16 prover MSMs via crates/epoch-snark/src/api/prover.rs:78.  Parity = equality of the affine-normalised
group element (bit-exact; SURVEY.md §7 "Bit-exactness must be defined on canonical forms")."""
import os
import numpy as np
import pytest
import torch
from oracle.py import ecc
from oracle import cpu_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _affine(out, kind):
    return co.jac_to_affine(out, kind)


def _gen_points_gpu(gpu, group, n, seed, gen_xy_limbs, words_per_point):
    t = torch.empty(n * words_per_point, dtype=torch.int64, device="cuda")
    gpu.gen_points_dev(group, t.data_ptr(), n, seed, gen_xy_limbs)
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 256, 1024])
def test_g1_small_vs_python(gpu, n):
    """Small sizes incl. edge scalars (0, 1, r-1, 2^64, 136-bit), an infinity base and a repeated base;
    expected value from the Python big-int definition (sum of scalar muls)."""
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 100 + n)
    sc = H.seeded_scalars(n, 200 + n, ecc.R377)
    if n >= 33:
        pts[7] = None
        pts[12] = pts[13]
        sc[12] = sc[13]          # identical (point, scalar) pairs land in the same bucket: doubling branch
        pts[20] = ecc.E1_377.neg(pts[21])
        sc[20] = sc[21]          # P and -P with equal digits: cancellation branch
    xy, inf = co.pack_g1_377(pts)
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=4), "g1_377")
    if n <= 256:
        assert exp == ecc.E1_377.msm(pts, sc)
    got = _affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377")
    assert got == exp


def test_g1_empty_and_all_zero(gpu):
    out = gpu.msm("bls12_377_g1", np.zeros((0, 12), dtype=np.uint64), None, np.zeros((0, 4), dtype=np.uint64))
    assert _affine(out, "g1_377") is None
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, 40, 5)
    xy, inf = co.pack_g1_377(pts)
    out = gpu.msm("bls12_377_g1", xy, inf, np.zeros((40, 4), dtype=np.uint64))
    assert _affine(out, "g1_377") is None
    # all scalars 1 == plain aggregate (Signature::aggregate, crates/bls-crypto/src/bls/signature.rs:61-67)
    ones = np.zeros((40, 4), dtype=np.uint64)
    ones[:, 0] = 1
    agg = None
    for P in pts:
        agg = ecc.E1_377.add(agg, P)
    assert _affine(gpu.msm("bls12_377_g1", xy, inf, ones), "g1_377") == agg


def test_g1_skewed_scalars(gpu):
    """Bucket skew: every scalar equal (one bucket per window gets all points) and witness-like 0/1-heavy scalars."""
    n = 512
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 77)
    xy, inf = co.pack_g1_377(pts)
    k = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789AB % ecc.R377
    s = H.scalars_np([k] * n, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=4), "g1_377")
    assert _affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377") == exp
    rng = ecc.SplitMix64(9)
    sc = [(0 if (rng.next() % 10) < 4 else 1 if (rng.next() % 10) < 5 else ecc.random_scalar(rng, ecc.R377)) for _ in range(n)]
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=4), "g1_377")
    assert _affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377") == exp


def test_g1_bucket_runs_of_equal_and_opposite_points(gpu):
    """Every scalar equal and the bases in runs of copies and of (P, -P) pairs: each window's one bucket run starts with two equal
    points (the affine + affine start of k_accumulate takes its doubling branch), continues with cancelling pairs (its identity
    branch, then mixed additions onto the identity) - in every order the sort may leave them, since all keys are equal."""
    rng = ecc.SplitMix64(31)
    P = ecc.E1_377.mul(ecc.G1_377, rng.next())
    Q = ecc.E1_377.mul(ecc.G1_377, rng.next())
    k = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F00F1E2D3C4B5A69788796A5B4C3D2E1 % ecc.R377
    for pts in ([P, P], [P, ecc.E1_377.neg(P)], [P, P, P], [P, ecc.E1_377.neg(P), Q], [P] * 5 + [ecc.E1_377.neg(P)] * 5 + [Q, Q],
                [ecc.E1_377.neg(P), P] * 40 + [Q] * 3, [P] * 97):
        xy, inf = co.pack_g1_377(pts)
        s = H.scalars_np([k] * len(pts), 4)
        exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=2), "g1_377")
        assert _affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377") == exp
        assert _affine(gpu.msm("bls12_377_g1", xy, inf, s, subgroup=True), "g1_377") == exp


@pytest.mark.parametrize("logn", [16, 20])
def test_g1_heavy_skew_large(gpu, logn):
    """Skew at sizes where runs are cut into many pieces (k_combine_mid / k_combine_big) and one region of the two-level sort
    spans many tiles (k_tile_count / k_tile_sort; 2^20 is the 16-bit-window, 128-bin configuration of the headline): (a) every
    scalar equal -> one bucket per window holds all points, (b) witness-like: 40% zero, 30% one, rest uniform."""
    n = 1 << logn
    gen, _ = co.pack_g1_377([ecc.G1_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g1", n, 0xABCD, gen.reshape(-1), 12)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 12)
    k = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789AB % ecc.R377
    sc = np.tile(co.ints_to_limbs([k], 4), (n, 1))
    exp = co.jac_to_affine(co.msm("bls12_377_g1", h_bases, None, sc, threads=8), "g1_377")
    assert _affine(gpu.msm("bls12_377_g1", h_bases, None, sc), "g1_377") == exp
    rng = np.random.default_rng(17)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    kind = rng.integers(0, 10, size=n)
    sc[kind < 4] = 0
    sc[(kind >= 4) & (kind < 7)] = np.array([1, 0, 0, 0], dtype=np.uint64)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", h_bases, None, sc, threads=8), "g1_377")
    assert _affine(gpu.msm("bls12_377_g1", h_bases, None, sc), "g1_377") == exp


@pytest.mark.parametrize("c", [4, 7, 11, 13, 16])
def test_g1_window_sizes(gpu, c):
    n = 700
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 31)
    sc = H.seeded_scalars(n, 32, ecc.R377)
    xy, inf = co.pack_g1_377(pts)
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", xy, inf, s, threads=4), "g1_377")
    gpu.set_window_bits("bls12_377_g1", c)
    try:
        assert _affine(gpu.msm("bls12_377_g1", xy, inf, s), "g1_377") == exp
        assert gpu.msm_timings("bls12_377_g1")["window_bits"] == c
    finally:
        gpu.set_window_bits("bls12_377_g1", 0)


@pytest.mark.parametrize("logn", [14, 17, 20])
def test_g1_large_device_resident(gpu, logn):
    """BASELINE config 2 shape: bases generated on the device (P_i = k_i*G), uniform scalars < r, inputs resident in
    HBM; the C++ oracle (arkworks Pippenger restatement, all host threads) runs on the same buffers.  Also checks the
    generator kernel against the oracle on a sample and the linearity property MSM(2s) == 2*MSM(s)."""
    n = 1 << logn
    gen, _ = co.pack_g1_377([ecc.G1_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g1", n, 0x5EED0002, gen.reshape(-1), 12)
    rng = np.random.default_rng(0x5EED0001)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)   # < 2^252 < r: uniform 252-bit scalars
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 12)
    # sample-check the generated bases: P_i = (splitmix64(seed, i) | 1) * G
    for i in (0, 1, n // 2, n - 1):
        k = H.splitmix64_at(0x5EED0002, i) | 1
        exp_pt = ecc.E1_377.mul(ecc.G1_377, k)
        got_pt = tuple(co.from_mont(h_bases[i].reshape(2, 6), ecc.Q377))
        assert got_pt == exp_pt
    threads = max(1, min(32, co.lib().orc_hardware_threads()))
    exp = co.msm("bls12_377_g1", h_bases, None, sc, threads=threads)
    assert _affine(out, "g1_377") == co.jac_to_affine(exp, "g1_377")
    if logn <= 17:
        sc2 = sc.copy()
        sc2[:, 3] &= np.uint64((1 << 59) - 1)
        d2 = torch.from_numpy(sc2.view(np.int64)).cuda()
        a = _affine(gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d2.data_ptr(), n), "g1_377")
        # doubled scalars: shift left by one bit
        carry = (sc2 >> np.uint64(63))
        sc2d = (sc2 << np.uint64(1))
        sc2d[:, 1:] |= carry[:, :-1]
        d3 = torch.from_numpy(sc2d.view(np.int64)).cuda()
        b = _affine(gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d3.data_ptr(), n), "g1_377")
        assert b == ecc.E1_377.add(a, a)


def test_g1_16_bit_windows_edge_scalars_and_odd_size(gpu):
    """The 16-bit configuration runs windows of mixed width (14 x 16 + 2 x 15 bits, csrc/msm.h k_digits): (a) scalars at the edges of
    the recoding - r - 1, r - 2, 2^252, the window boundaries 2^224, 2^239, 2^240 and their neighbours, runs of ones that carry
    through every window - against the big-integer definition; (b) an odd size above 2^19 (the size from which the configuration is
    chosen: ragged last blocks and tiles of the two-level sort), with r - 1 - i among uniform scalars, against the C++ oracle."""
    r = ecc.R377
    edge = [r - 1, r - 2, 1, 2, 1 << 252, (1 << 252) - 1, (1 << 252) + 1, 1 << 224, (1 << 224) - 1, 1 << 239, (1 << 239) - 1,
            1 << 240, (1 << 240) - 1, (1 << 16) - 1, 1 << 15, (1 << 15) + 1, (1 << 252) - (1 << 15), r >> 1, (r >> 1) + 1,
            int("8000" * 15, 16) >> 3, int("7fff" * 15, 16), int("8001" * 15, 16) >> 3]
    edge = [k % r for k in edge]
    n = len(edge)
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 1601)
    xy, inf = co.pack_g1_377(pts)
    exp = None
    for P, k in zip(pts, edge):
        exp = ecc.E1_377.add(exp, ecc.E1_377.mul(P, k))
    gpu.set_window_bits("bls12_377_g1", 16)
    try:
        assert _affine(gpu.msm("bls12_377_g1", xy, inf, H.scalars_np(edge, 4)), "g1_377") == exp
        for k, P in zip(edge, pts):      # one term at a time: a wrong digit cannot cancel against another term
            pxy, pinf = co.pack_g1_377([P])
            assert _affine(gpu.msm("bls12_377_g1", pxy, pinf, H.scalars_np([k], 4)), "g1_377") == ecc.E1_377.mul(P, k), hex(k)
    finally:
        gpu.set_window_bits("bls12_377_g1", 0)
    n = (1 << 19) + 12345
    gen, _ = co.pack_g1_377([ecc.G1_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g1", n, 0x5EED0019, gen.reshape(-1), 12)
    rng = np.random.default_rng(1919)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    sc[:2000] = co.ints_to_limbs([r - 1 - i for i in range(2000)], 4)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    assert gpu.msm_timings("bls12_377_g1")["window_bits"] == 16
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 12)
    exp = co.msm("bls12_377_g1", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g1_377") == co.jac_to_affine(exp, "g1_377")


def test_g1_two_to_22_device_resident(gpu):
    """Beyond the headline size (BASELINE config 5's G1 leg is 2^22): oracle comparison at 4M terms."""
    n = 1 << 22
    gen, _ = co.pack_g1_377([ecc.G1_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g1", n, 0x5EED0022, gen.reshape(-1), 12)
    rng = np.random.default_rng(22)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 12)
    exp = co.msm("bls12_377_g1", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g1_377") == co.jac_to_affine(exp, "g1_377")


def test_bw6_761_two_to_17_device_resident(gpu, golden):
    """BASELINE config 4 shape (Groth16 prover MSM over BW6-761 G1) at 2^17 terms per GPU, uniform 376-bit scalars."""
    from oracle.py import epoch as ep
    vk = ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))
    n = 1 << 17
    gen, _ = co.pack_761([vk["alpha_g1"]])
    bases = _gen_points_gpu(gpu, "bw6_761_g1", n, 0x5EED0761, gen.reshape(-1), 24)
    rng = np.random.default_rng(761)
    sc = rng.integers(0, 1 << 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 6), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 5] &= np.uint64((1 << 56) - 1)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    exp = co.msm("bw6_761_g1", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "761") == co.jac_to_affine(exp, "761")


@pytest.mark.parametrize("n", [1, 33, 300])
def test_g2_vs_oracle(gpu, n):
    pts = H.seeded_points(ecc.E2_377, ecc.G2_377, n, 300 + n)
    sc = H.seeded_scalars(n, 400 + n, ecc.R377)
    if n >= 33:
        pts[3] = None
        pts[8] = pts[9]
        sc[8] = sc[9]
    xy, inf = co.pack_g2_377(pts)
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g2", xy, inf, s, threads=4), "g2_377")
    assert _affine(gpu.msm("bls12_377_g2", xy, inf, s), "g2_377") == exp


def test_g2_large_device_resident(gpu):
    n = 1 << 14
    gen, _ = co.pack_g2_377([ecc.G2_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g2", n, 0x5EED0003, gen.reshape(-1), 24)
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g2", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    k = H.splitmix64_at(0x5EED0003, 5) | 1
    v = co.from_mont(h_bases[5].reshape(4, 6), ecc.Q377)
    assert ((v[0], v[1]), (v[2], v[3])) == ecc.E2_377.mul(ecc.G2_377, k)
    exp = co.msm("bls12_377_g2", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g2_377") == co.jac_to_affine(exp, "g2_377")


@pytest.mark.parametrize("log_n,seed", [(17, 1), (17, 2), (17, 3), (20, 1), (20, 2), (20, 3)])
def test_g2_plain_entry_two_to_17_and_20(gpu, log_n, seed):
    """the PLAIN G2 entry point (msm_bls12_377_g2_dev, no endomorphism) at 2^17 and 2^20 terms, three seeds each - the
    sizes where the round-3 signed form of the Fq2 R t - Y1 PPP pass produced wrong sums while every smaller test stayed green, and
    which -m gpu did not cover (2^14 plain, 2^17 only through the _subgroup entry, 2^22 through config 5).  Uniform 253-bit scalars with
    r - 1, 2^252 and runs of equal / opposite bases mixed in."""
    n = 1 << log_n
    gen, _ = co.pack_g2_377([ecc.G2_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g2", n, 0x5EED2000 + 97 * seed + log_n, gen.reshape(-1), 24)
    rng = np.random.default_rng(1000 * log_n + seed)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)                       # below 2^252 < r
    sc[:4] = H.scalars_np([ecc.R377 - 1, 1 << 252, 0, 1], 4)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 24).copy()
    h_bases[100:108] = h_bases[100]                            # equal bases with equal scalars: the doubling branch inside a bucket
    sc[100:108] = sc[100]
    h_bases[200] = h_bases[201]; h_bases[200, 12:18] = co.to_mont([(ecc.Q377 - v) % ecc.Q377 for v in co.from_mont(h_bases[201, 12:18], ecc.Q377)], ecc.Q377)[0]
    h_bases[200, 18:24] = co.to_mont([(ecc.Q377 - v) % ecc.Q377 for v in co.from_mont(h_bases[201, 18:24], ecc.Q377)], ecc.Q377)[0]
    sc[200] = sc[201]                                          # opposite bases with equal scalars: the cancellation branch
    d_b = torch.from_numpy(h_bases.view(np.int64)).cuda()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g2", d_b.data_ptr(), 0, d_sc.data_ptr(), n)
    exp = co.msm("bls12_377_g2", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g2_377") == co.jac_to_affine(exp, "g2_377")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bw6_761_g2_two_to_17(gpu, golden, seed):
    """BW6-761 G2 (the prover's b_g2_query) at 2^17 terms, three seeds: -m gpu checked this group at n <= 2048 only."""
    from oracle.py import epoch as ep
    vk = ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))
    n = 1 << 17
    gen, _ = co.pack_761([vk["beta_g2"]])
    bases = _gen_points_gpu(gpu, "bw6_761_g2", n, 0x5EED7612 + seed, gen.reshape(-1), 24)
    rng = np.random.default_rng(7610 + seed)
    sc = rng.integers(0, 1 << 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 6), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 5] &= np.uint64((1 << 56) - 1)
    sc[:3] = H.scalars_np([ecc.R761 - 1, 1 << 376, 1], 6)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bw6_761_g2", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    exp = co.msm("bw6_761_g2", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "761") == co.jac_to_affine(exp, "761")


@pytest.mark.parametrize("n", [1, 40, 2048])
def test_bw6_761_vs_oracle(gpu, golden, n):
    """BASELINE config 4 shape (Groth16 prover MSM over BW6-761 G1/G2) at oracle-checkable sizes; base points derived
    from the reference's Groth16 verifying key (the only BW6-761 points in the tree)."""
    from oracle.py import epoch as ep
    vk = ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))
    for grp, G, cur in (("bw6_761_g1", vk["alpha_g1"], ecc.E1_761), ("bw6_761_g2", vk["beta_g2"], ecc.E2_761)):
        if n <= 40:
            rng = ecc.SplitMix64(n)
            pts = [cur.mul(G, rng.next() | 1) for _ in range(n)]
            if n == 40:
                pts[3] = None
            xy, inf = co.pack_761(pts)
        else:
            gen, _ = co.pack_761([G])
            t = _gen_points_gpu(gpu, grp, n, 0x5EED0004, gen.reshape(-1), 24)
            xy, inf = t.cpu().numpy().view(np.uint64).reshape(n, 24), None
        sc = H.seeded_scalars(n, 500 + n, ecc.R761)
        s = H.scalars_np(sc, 6)
        exp = co.jac_to_affine(co.msm(grp, xy, inf, s, threads=8), "761")
        assert _affine(gpu.msm(grp, xy, inf, s), "761") == exp


def test_bw6_761_16_bit_windows_edge_scalars(gpu, golden):
    """BW6-761's 16-bit configuration: 18 x 16 + 6 x 15 = 378 bits.  Edge scalars of the 377-bit scalar field (r - 1, powers of two at
    the wide / narrow boundary 2^288 and at the narrow windows' edges, carries through every window), each alone and all together,
    against the big-integer definition; both groups."""
    from oracle.py import epoch as ep
    vk = ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))
    r = ecc.R761
    edge = [r - 1, r - 2, 1, 1 << 376, (1 << 376) - 1, 1 << 288, (1 << 288) - 1, 1 << 303, (1 << 303) - 1, 1 << 363, (1 << 363) - 1,
            (1 << 376) - (1 << 15), r >> 1, int("8000" * 23, 16), int("7fff" * 23, 16), int("8001" * 23, 16)]
    edge = [k % r for k in edge]
    for grp, G, cur in (("bw6_761_g1", vk["alpha_g1"], ecc.E1_761), ("bw6_761_g2", vk["beta_g2"], ecc.E2_761)):
        rng = ecc.SplitMix64(761)
        pts = [cur.mul(G, rng.next() | 1) for _ in edge]
        xy, inf = co.pack_761(pts)
        exp = None
        for P, k in zip(pts, edge):
            exp = cur.add(exp, cur.mul(P, k))
        gpu.set_window_bits(grp, 16)
        try:
            assert _affine(gpu.msm(grp, xy, inf, H.scalars_np(edge, 6)), "761") == exp
            for k, P in zip(edge[:8], pts):
                pxy, pinf = co.pack_761([P])
                assert _affine(gpu.msm(grp, pxy, pinf, H.scalars_np([k], 6)), "761") == cur.mul(P, k), hex(k)
        finally:
            gpu.set_window_bits(grp, 0)


def test_gpu_reproduces_the_frozen_msm_fixtures(gpu):
    """tests/golden/msm_fixtures.json: results frozen from the big-int definition (tests/golden/make_msm_fixtures.py) - the GPU
    path against committed data, not only against a checker computed in the same run."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_fixtures.json")) as f:
        fx = json.load(f)
    for n, want in fx["g1"].items():
        n = int(n)
        pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 100 + n)
        sc = H.seeded_scalars(n, 200 + n, ecc.R377)
        xy, inf = co.pack_g1_377(pts)
        got = _affine(gpu.msm("bls12_377_g1", xy, inf, H.scalars_np(sc, 4)), "g1_377")
        assert got == (int(want[0], 16), int(want[1], 16)), n
    for n, want in fx["g2"].items():
        n = int(n)
        pts = H.seeded_points(ecc.E2_377, ecc.G2_377, n, 100 + n)
        sc = H.seeded_scalars(n, 200 + n, ecc.R377)
        xy, inf = co.pack_g2_377(pts)
        got = _affine(gpu.msm("bls12_377_g2", xy, inf, H.scalars_np(sc, 4)), "g2_377")
        assert got == ((int(want[0][0], 16), int(want[0][1], 16)), (int(want[1][0], 16), int(want[1][1], 16))), n


# ------------------------------------------------------------------------------------------------ msm_bls12_377_g1_subgroup (GLV split)
@pytest.mark.parametrize("n", [1, 2, 33, 256, 1024, 5000])
def test_g1_subgroup_entry_small_vs_python(gpu, n):
    """msm_bls12_377_g1_subgroup (bases vouched to lie in G1: Signature::batch's inputs, crates/bls-crypto/src/bls/signature.rs:70-89):
    the GLV split k = k0 + k1 x^2, [x^2]P = (beta x, -y) (csrc/msm.h k_glv_expand, gls.h) against the big-integer definition - edge
    scalars on both sides of x^2 and of the halves' window borders, 0, 1, r - 1, an identity base, repeated and opposite points."""
    X2 = 0x8508C00000000001 ** 2
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 7100 + n)
    sc = H.seeded_scalars(n, 7200 + n, ecc.R377)
    edge = [0, 1, ecc.R377 - 1, X2 - 1, X2, X2 + 1, 2 * X2 - 1, (1 << 127) - 1, 1 << 127, (1 << 126), (X2 - 1) + X2 * ((1 << 126) + 5), (1 << 16) - 1, 1 << 15,
            X2 * ((1 << 112) - 1), (1 << 252) + 1]
    for i, k in enumerate(edge):
        if i < n:
            sc[i] = k % ecc.R377
    if n >= 33:
        pts[17] = None
        pts[22] = pts[23]; sc[22] = sc[23]
        pts[25] = ecc.E1_377.neg(pts[26]); sc[25] = sc[26]
    xy, inf = co.pack_g1_377(pts)
    s = H.scalars_np(sc, 4)
    exp = None
    for P, k in zip(pts, sc):
        exp = ecc.E1_377.add(exp, ecc.E1_377.mul(P, k))
    assert _affine(gpu.msm("bls12_377_g1", xy, inf, s, subgroup=True), "g1_377") == exp
    if n <= 33:
        for P, k in zip(pts, sc):                  # one term at a time: a wrong half cannot cancel against another term
            if P is None:
                continue
            pxy, pinf = co.pack_g1_377([P])
            assert _affine(gpu.msm("bls12_377_g1", pxy, pinf, H.scalars_np([k], 4), subgroup=True), "g1_377") == ecc.E1_377.mul(P, k), hex(k)


@pytest.mark.parametrize("logn", [14, 17, 20])
def test_g1_subgroup_entry_large_device_resident(gpu, logn):
    """the same entry point at the bench's sizes, device-resident, against the C++ oracle and against the plain entry point"""
    n = 1 << logn
    gen, _ = co.pack_g1_377([ecc.G1_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g1", n, 0x5EED0002, gen.reshape(-1), 12)
    rng = np.random.default_rng(0x5EED0041)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    sc[:1000] = co.ints_to_limbs([ecc.R377 - 1 - i for i in range(1000)], 4)
    sc[1000:1004, 3] |= np.uint64(0xE000000000000000)      # bits from Fr::MODULUS_BITS up: ignored by both entry points (and by ark-ec's windows)
    # the split's edge scalars AT A SIZE WHERE THE SPLIT RUNS (n >= 2^14; the small test above takes the plain path - ADVICE r3): both
    # sides of x^2 and of the halves' window borders, 0, 1, and their neighbours, spread over the index range
    X2 = 0x8508C00000000001 ** 2
    edge = [0, 1, 2, X2 - 1, X2, X2 + 1, 2 * X2 - 1, 2 * X2, (1 << 127) - 1, 1 << 127, (1 << 127) + 1, 1 << 126, (X2 - 1) + X2 * ((1 << 126) + 5),
            (1 << 16) - 1, 1 << 16, 1 << 15, (1 << 15) - 1, X2 * ((1 << 112) - 1), X2 * ((1 << 126) - 1) + (X2 - 1), (1 << 252) + 1, ecc.R377 - X2, ecc.R377 - X2 - 1]
    edge = [k % ecc.R377 for k in edge]
    at = [1004 + 37 * i for i in range(len(edge))] + [n - 1 - 53 * i for i in range(len(edge))]
    sc[at] = co.ints_to_limbs(edge + edge, 4)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n, subgroup=True)
    tm = gpu.msm_timings("bls12_377_g1")
    assert tm["windows"] <= 12                      # the split halves the window count (8 x 16 bits at 2^20)
    plain = gpu.msm_dev("bls12_377_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    assert _affine(out, "g1_377") == _affine(plain, "g1_377")
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 12)
    sc[1000:1004, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)      # what both entry points computed with
    exp = co.msm("bls12_377_g1", h_bases, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g1_377") == co.jac_to_affine(exp, "g1_377")


@pytest.mark.parametrize("n", [1, 40, 700])
def test_g2_subgroup_entry_small_vs_oracle(gpu, n):
    """msm_bls12_377_g2_subgroup (bases vouched to lie in G2: PublicKey::batch's inputs, crates/bls-crypto/src/bls/public.rs:47-65): below
    2^14 terms the plain path runs; same sums as the oracle with identities and repeated points inside."""
    pts = H.seeded_points(ecc.E2_377, ecc.G2_377, n, 8100 + n)
    sc = H.seeded_scalars(n, 8200 + n, ecc.R377)
    if n >= 40:
        pts[3] = None
        pts[8] = pts[9]; sc[8] = sc[9]
    xy, inf = co.pack_g2_377(pts)
    s = H.scalars_np(sc, 4)
    exp = co.jac_to_affine(co.msm("bls12_377_g2", xy, inf, s, threads=4), "g2_377")
    assert _affine(gpu.msm("bls12_377_g2", xy, inf, s, subgroup=True), "g2_377") == exp


@pytest.mark.parametrize("logn", [14, 17])
def test_g2_subgroup_entry_glv_split_device_resident(gpu, logn):
    """from 2^14 terms the split k = k0 + k1 x^2, [x^2]P = psi^2(P) runs (8 windows of 16 bits over 2n points): against the C++ oracle and
    the plain entry point, with r - 1 - i, 0, 1, x^2 and x^2 - 1 among uniform scalars and identity-flagged bases"""
    n = 1 << logn
    gen, _ = co.pack_g2_377([ecc.G2_377])
    bases = _gen_points_gpu(gpu, "bls12_377_g2", n, 0x5EED0052, gen.reshape(-1), 24)
    rng = np.random.default_rng(0x5EED0051)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    sc ^= rng.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    X2 = 0x8508C00000000001 ** 2
    sc[:200] = co.ints_to_limbs([ecc.R377 - 1 - i for i in range(196)] + [0, 1, X2, X2 - 1], 4)
    inf = np.zeros(n, dtype=np.uint8); inf[[5, 77, n - 1]] = 1
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    d_inf = torch.from_numpy(inf).cuda()
    out = gpu.msm_dev("bls12_377_g2", bases.data_ptr(), d_inf.data_ptr(), d_sc.data_ptr(), n, subgroup=True)
    assert gpu.msm_timings("bls12_377_g2")["windows"] == 8
    plain = gpu.msm_dev("bls12_377_g2", bases.data_ptr(), d_inf.data_ptr(), d_sc.data_ptr(), n)
    assert _affine(out, "g2_377") == _affine(plain, "g2_377")
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    exp = co.msm("bls12_377_g2", h_bases, inf, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert _affine(out, "g2_377") == co.jac_to_affine(exp, "g2_377")


def test_shipped_g2_accumulate_kernel_equals_the_host_replay(gpu):
    """The regression guard that came out of the round-3 "signed pass" finding: the library's k_accumulate<G2_377>
    - 256 VGPRs + 220 AGPRs, the kernel whose signed-pass instantiation computes wrong sums - in its SHIPPED (unsigned) instantiation,
    on 16384 bucket runs of 24 random signed G2 points, against the host replay of the same templates for 2048 of the runs: every
    partial sum limb for limb (tools/repro_acc, built by the Makefile).  The same tool runs the signed instantiation beside it; how
    many of ITS runs differ depends on the compiler's register allocation and is printed, not asserted."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "celo-bls-snark-rs_amd", "build", "repro_acc")
    assert os.path.exists(exe), "tools/repro_acc not built (make -C celo-bls-snark-rs_amd/csrc)"
    r = subprocess.run([exe, "14", "24", "1", "2048"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-600:])
    host = [ln for ln in r.stdout.splitlines() if ln.startswith("host replay")]
    assert host and host[-1].endswith(": 0 differ"), r.stdout[-2000:] + r.stderr[-2000:]
    assert r.returncode in (0, 2)          # 2: the signed instantiation differs somewhere (the finding itself), 4 would be the shipped kernel


@pytest.mark.parametrize("group", ["bls12_377_g1", "bls12_377_g2", "bw6_761_g1"])
@pytest.mark.parametrize("chunked", [False, True])
def test_library_accumulate_kernels_equal_the_host_replay(gpu, golden, group, chunked):
    """The kernels the LIBRARY launches (not a tool's own compilation of the template): k_accumulate<G> and the host-pointer pipeline's
    k_accumulate_chunk<G>, every group, 16384 runs of 24 (+ 24 carried-on) random signed points with equal / opposite pairs at the head of some
    runs, the first 2048 partial sums against the host replay of the same formulas limb for limb (celo_amd_selftest_accumulate; round 5: the
    guard of round 4 covered one group in a side build).  Also what the build says about these kernels' registers: no SGPR spills beyond
    a handful of exec masks (tests/test_abi_symbols.py)."""
    from oracle.py import epoch as ep
    if group == "bls12_377_g1":
        gen = co.pack_g1_377([ecc.G1_377])[0]
    elif group == "bls12_377_g2":
        gen = co.pack_g2_377([ecc.G2_377])[0]
    else:
        gen = co.pack_761([ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))["alpha_g1"]])[0]
    for seed in (1, 2):
        assert gpu.selftest_accumulate(group, gen.reshape(-1), runs=16384, length=24, seed=seed, check=2048, chunked=chunked) == 0
    assert gpu.selftest_accumulate(group, gen.reshape(-1), runs=300, length=1, seed=3, check=300, chunked=chunked) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_g2_accumulation_on_lane_pairs_matches_the_oracle(variant):
    """CELO_G2_PAIR=0 | 1 | 2 (read once per process: a child process): the one-lane kernel, and k_accumulate_pair<G2_377> - the two halves of
    every Fq2 value on two adjacent lanes, pair-uniform branches - in its two forms (2 is the library's default).  G2 MSMs at 2^10 .. 2^17 terms incl. the branches of the
    mixed addition a bucket run can take: the same point twice in a bucket (doubling), a point and its negative (cancellation, then a
    restart from the identity), all scalars equal (one long run per window), infinity flags; and a chained Batch::verify."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import numpy as np, torch
        from oracle.py import ecc
        from oracle import cpu_oracle as co
        from celo_bls_snark_rs_amd import ffi, synthetic as syn
        ffi.init(0)
        g2, _ = co.pack_g2_377([ecc.G2_377])
        T = max(1, min(32, co.lib().orc_hardware_threads()))
        rng = np.random.default_rng(11)
        def pts(n, seed):
            t = torch.empty(n * 24, dtype=torch.int64, device="cuda")
            ffi.gen_points_dev("bls12_377_g2", t.data_ptr(), n, seed, g2.reshape(-1))
            torch.cuda.synchronize()
            return t.cpu().numpy().view(np.uint64).reshape(n, 24)
        def scal(n, seed):
            r = np.random.default_rng(seed)
            sc = r.integers(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
            sc[:, 3] &= np.uint64((1 << 60) - 1)
            return sc
        for n, seed in ((1 << 10, 1), (5000, 2), ((1 << 17) + 3, 3)):
            xy = pts(n, seed); sc = scal(n, seed + 50)
            inf = np.zeros(n, dtype=np.uint8); inf[[1, n - 1]] = 1
            # equal points with equal scalars (doubling inside a bucket), opposite points with equal scalars (cancellation)
            xy[7] = xy[3]; sc[7] = sc[3]
            P = tuple(co.from_mont(xy[9].reshape(4, 6), ecc.Q377)); P = ((P[0], P[1]), (P[2], P[3]))
            xy[11] = co.pack_g2_377([ecc.E2_377.neg(P)])[0][0]; sc[11] = sc[9]
            xy[12] = xy[9]; sc[12] = sc[9]                                    # ... and the run goes on after the cancellation
            assert co.jac_to_affine(ffi.msm("bls12_377_g2", xy, inf, sc), "g2_377") == co.jac_to_affine(co.msm("bls12_377_g2", xy, inf, sc, threads=T), "g2_377"), n
            sc[:] = sc[5]                                                      # one scalar for all: every window is one long run
            xy[100:200] = xy[100]                                              # ... with a stretch of one repeated point in it
            assert co.jac_to_affine(ffi.msm("bls12_377_g2", xy, None, sc), "g2_377") == co.jac_to_affine(co.msm("bls12_377_g2", xy, None, sc, threads=T), "g2_377"), n
        w = syn.valid_batches(6, 40, 99, [2])
        ex = syn.batch_exponents(240, 100)
        d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
        ok = ffi.batch_verify_dev(w["pk"].data_ptr(), w["sig"].data_ptr(), d_ex.data_ptr(), w["offsets"], w["hash"].data_ptr(), syn.neg_g2_limbs())
        assert ok.tolist() == [1, 1, 0, 1, 1, 1]
        print("PAIR-OK")
    ''')
    env = dict(os.environ, CELO_G2_PAIR=str(variant))
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=480)
    assert r.returncode == 0 and "PAIR-OK" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
