"""GPU (-m gpu): the batched-affine pre-levels of the BW6-761 bucket accumulation (csrc/msm_ba.h) against the oracle on the inputs that reach
its special pairs - equal points in one bucket (the doubling branch: denominator 2 y), opposite points (the identity as a level RESULT), the
identity as an OPERAND of a later level, odd runs (the copied leftover), runs of one - and on uniform / witness-like scalars at sizes where
the tree is three levels deep.  Reference call site: ark_groth16's prover MSMs via crates/epoch-snark/src/api/prover.rs:78.  Parity = equality
of the affine-normalised group element with the CPU port's (bit-exact)."""
import numpy as np
import pytest
import torch
from oracle.py import ecc
from oracle import cpu_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _batched_affine_on(gpu):
    """The pre-levels are off by default (they measure level with the XYZZ chain: profiles/r6_ba_ab.txt); these tests switch them on."""
    gpu.set_batched_affine(1)
    yield
    gpu.set_batched_affine(-1)


def _vk_point(golden):
    from oracle.py import epoch as ep
    return ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))["alpha_g1"]


def test_ba_special_pairs_in_one_bucket(gpu, golden):
    """Every scalar equal: each window has ONE bucket run, cut into pieces; its pairs meet every branch of BaOps::classify."""
    cur = ecc.E1_761
    G = _vk_point(golden)
    rng = ecc.SplitMix64(7611)
    P, Q, R = (cur.mul(G, rng.next() | 1) for _ in range(3))
    nP, nQ = cur.neg(P), cur.neg(Q)
    k = 0x1F1E2D3C4B5A69788796A5B4C3D2E1F00F1E2D3C4B5A69788796A5B4C3D2E1F00F1E2D3C4B5A69788796A5B4C3D2E1 % ecc.R761
    shapes = [[P, P], [P, nP], [P, P, P], [P, nP, Q], [P, nP, Q, nQ], [P, nP, Q, nQ, R], [P] * 4, [P] * 8 + [Q], [P, P, nP, nP] * 5 + [R],
              [nP, P] * 40 + [Q] * 3, [P] * 97, [P, Q] * 64 + [nQ, nP] * 64, [P, Q, R] * 50, [P, nP] * 3 + [Q, Q] * 5 + [R] * 7]
    for pts in shapes:
        xy, inf = co.pack_761(pts)
        s = H.scalars_np([k] * len(pts), 6)
        exp = co.jac_to_affine(co.msm("bw6_761_g1", xy, inf, s, threads=2), "761")
        assert co.jac_to_affine(gpu.msm("bw6_761_g1", xy, inf, s), "761") == exp, len(pts)


@pytest.mark.parametrize("n", [1 << 12, (1 << 14) + 77])
def test_ba_repeated_and_negated_bases_random_scalars(gpu, golden, n):
    """64 distinct points and their negatives tiled over n terms with scalars drawn from a set of 16: (point, scalar) collisions put equal and
    opposite points into the same buckets of every window, between generic pairs."""
    cur = ecc.E1_761
    G = _vk_point(golden)
    rng = ecc.SplitMix64(7612 + n)
    base = [cur.mul(G, rng.next() | 1) for _ in range(64)]
    base += [cur.neg(p) for p in base[:32]]
    xy_b, _ = co.pack_761(base)
    pick = np.random.default_rng(n).integers(0, len(base), size=n)
    xy = xy_b[pick]
    ks = [ecc.random_scalar(rng, ecc.R761) for _ in range(16)]
    sc = [ks[i] for i in np.random.default_rng(n + 1).integers(0, 16, size=n)]
    s = H.scalars_np(sc, 6)
    exp = co.jac_to_affine(co.msm("bw6_761_g1", xy, None, s, threads=8), "761")
    assert co.jac_to_affine(gpu.msm("bw6_761_g1", xy, None, s), "761") == exp


@pytest.mark.parametrize("kind", ["uniform", "witness"])
def test_ba_two_to_18_device_resident(gpu, golden, kind):
    """2^18 terms resident in HBM (mean bucket run 16 at c = 15: three tree levels, then the XYZZ chain), uniform and witness-like scalars (about
    60 % zeros and ones: the bucket of digit 1 in window 0 is ~50 000 points long and is cut into pieces)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 1 << 18
    bases = syn.device_points("bw6_761_g1", n, 0x5EED0BA0)
    sc = syn.uniform_scalars("bw6_761_g1", n, 0xBA1) if kind == "uniform" else syn.witness_like_scalars("bw6_761_g1", n, 0xBA2)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    h = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    exp = co.msm("bw6_761_g1", h, None, sc, threads=max(1, min(32, co.lib().orc_hardware_threads())))
    assert co.jac_to_affine(out, "761") == co.jac_to_affine(exp, "761")


def test_ba_on_and_off_agree(gpu, golden):
    """The switch itself: 2^16 resident terms with the pre-levels on and off give the same group element (and the timings show both ran)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    n = 1 << 16
    bases = syn.device_points("bw6_761_g1", n, 0x5EED0BA3)
    sc = syn.uniform_scalars("bw6_761_g1", n, 0xBA4)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    on = gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    gpu.set_batched_affine(0)
    off = gpu.msm_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n)
    assert co.jac_to_affine(on, "761") == co.jac_to_affine(off, "761")


def test_ba_bw6_761_g2_and_window_shards(gpu, golden):
    """The pre-levels serve every caller of the resident BW6-761 pipeline: the G2 group (same coordinate field, another b: the addition formulas
    do not use it and (0, 0) is off that curve too) at 2^14 terms, and the window shards of one 2^16-term G1 job (three shards joined by
    msm_*_join_windows: each shard's pieces are shorter, the tree has odd runs and runs of one)."""
    from oracle.py import epoch as ep
    from celo_bls_snark_rs_amd import synthetic as syn
    vk = ep.parse_vk(bytes.fromhex(golden["groth16_bw6_761"]["vk"]))
    n = 1 << 14
    gen, _ = co.pack_761([vk["beta_g2"]])
    t = torch.empty(n * 24, dtype=torch.int64, device="cuda")
    gpu.gen_points_dev("bw6_761_g2", t.data_ptr(), n, 0x5EED0BA5, gen.reshape(-1))
    sc = syn.uniform_scalars("bw6_761_g2", n, 0xBA6)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = gpu.msm_dev("bw6_761_g2", t.data_ptr(), 0, d_sc.data_ptr(), n)
    h = t.cpu().numpy().view(np.uint64).reshape(n, 24)
    assert co.jac_to_affine(out, "761") == co.jac_to_affine(co.msm("bw6_761_g2", h, None, sc, threads=8), "761")
    n = 1 << 16
    bases = syn.device_points("bw6_761_g1", n, 0x5EED0BA7)
    sc = syn.uniform_scalars("bw6_761_g1", n, 0xBA8)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    recs, bits = [], []
    for s in range(3):
        r, b = gpu.msm_window_shard_dev("bw6_761_g1", bases.data_ptr(), 0, d_sc.data_ptr(), n, s, 3)
        recs.append(r); bits.append(b)
    got = gpu.join_windows("bw6_761_g1", np.concatenate(recs), bits)
    h = bases.cpu().numpy().view(np.uint64).reshape(n, 24)
    assert co.jac_to_affine(got, "761") == co.jac_to_affine(co.msm("bw6_761_g1", h, None, sc, threads=8), "761")
