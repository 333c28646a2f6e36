"""GPU (-m gpu): the reference's only end-to-end pairing known-answer vector through the PRODUCT:
crates/bls-snark-sys/src/snark/mod.rs:52-119 (Groth16 over BW6-761) must ACCEPT on the HIP pairing kernels and REJECT when
tampered.  Decoding / hashing / input packing (plumbing) is done by the oracle-side Python; the two public
input scalar-muls go through the product's BW6-761 G1 MSM, the 4-pair product check through pairing_product_is_one_bw6_761."""
import numpy as np
import pytest
from oracle.py import ecc, epoch as ep
from oracle import cpu_oracle as co
from tests.test_oracle_golden import _groth16_setup

pytestmark = pytest.mark.gpu


def _pairs_via_product_msm(gpu, vk, pr, inputs):
    # acc = gamma_abc[0] + sum inputs[i] * gamma_abc[i+1]  — MSM with scalars (1, in_0, in_1) on the GPU
    pts = vk["gamma_abc_g1"]
    xy, inf = co.pack_761(pts)
    sc = co.ints_to_limbs([1] + list(inputs), 6)
    acc = co.jac_to_affine(gpu.msm("bw6_761_g1", xy, inf, sc), "761")
    return [(pr["a"], pr["b"]), (acc, ecc.E2_761.neg(vk["gamma_g2"])), (pr["c"], ecc.E2_761.neg(vk["delta_g2"])),
            (ecc.E1_761.neg(vk["alpha_g1"]), vk["beta_g2"])]


def test_groth16_reference_vector_accepts_on_gpu(gpu, golden):
    vk, pr, inputs = _groth16_setup(golden)
    pairs = _pairs_via_product_msm(gpu, vk, pr, inputs)
    assert pairs[1][0] == ep.groth16_pairs(vk, pr, inputs)[1][0]
    g1, i1 = co.pack_761([p for p, _ in pairs])
    g2, i2 = co.pack_761([q for _, q in pairs])
    assert gpu.pairing_product_is_one_bw6(g1, i1, g2, i2)
    # tamper: flip one bit of a public input; change the proof's C
    bad = _pairs_via_product_msm(gpu, vk, pr, [inputs[0] ^ 1, inputs[1]])
    g1b, _ = co.pack_761([p for p, _ in bad])
    assert not gpu.pairing_product_is_one_bw6(g1b, i1, g2, i2)
    bad_pr = dict(pr, c=ecc.E1_761.add(pr["c"], pr["c"]))
    pairs = _pairs_via_product_msm(gpu, vk, bad_pr, inputs)
    g1c, _ = co.pack_761([p for p, _ in pairs])
    assert not gpu.pairing_product_is_one_bw6(g1c, i1, g2, i2)


def test_bw6_gt_bit_exact(gpu, golden):
    vk, pr, _ = _groth16_setup(golden)
    g1, _ = co.pack_761([vk["alpha_g1"], pr["a"]])
    g2, _ = co.pack_761([vk["beta_g2"], pr["b"]])
    offs = np.array([0, 1, 2], dtype=np.uint32)
    gt = gpu.pairing_gt_bw6(g1, None, g2, None, offs)
    for i in range(2):
        assert np.array_equal(gt[i], co.pairing_product_761(g1[i:i + 1], None, g2[i:i + 1], None)[0])
    gt2 = gpu.pairing_gt_bw6(g1, None, g2, None, np.array([0, 2], dtype=np.uint32))
    assert np.array_equal(gt2[0], co.pairing_product_761(g1, None, g2, None)[0])


@pytest.mark.parametrize("k", [1, 3, 4, 7, 8])
def test_bw6_single_product_paths_bit_exact(gpu, golden, k):
    """ONE product of k pairs: k <= 7 takes the latency path (csrc/unit_pairing761_wide.hip: products of an Fq6 operation side by
    side in lane groups, the long Miller loop cut into a point wave and three iteration ranges, two-wave ladders), k = 8 the
    throughput kernels.  Miller value and GT value against the oracle's restatement of ark-ec's BW6 engine (what
    ark_groth16::verify_proof computes at crates/epoch-snark/src/api/verifier.rs:35), also with a pair at infinity."""
    vk, pr, _ = _groth16_setup(golden)
    P = [ecc.E1_761.mul(vk["alpha_g1"], 3 + 2 * j) if j % 2 else ecc.E1_761.mul(pr["a"], 5 + j) for j in range(k)]
    Q = [ecc.E2_761.mul(vk["beta_g2"], 7 + j) if j % 3 else ecc.E2_761.mul(pr["b"], 2 + j) for j in range(k)]
    offs = np.array([0, k], dtype=np.uint32)
    for drop in ((None, k // 2) if k > 1 else (None,)):
        g1, i1 = co.pack_761([None if j == drop else p for j, p in enumerate(P)])
        g2, i2 = co.pack_761(Q)
        want, one = co.pairing_product_761(g1, i1, g2, i2)
        gt = gpu.pairing_gt_bw6(g1, i1, g2, i2, offs)
        assert np.array_equal(gt[0], want)
        assert bool(gpu.pairing_product_is_one_bw6(g1, i1, g2, i2)) == bool(one)
