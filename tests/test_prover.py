"""Groth16 prover row: the witness map and create_proof_no_zk's group arithmetic
(crates/epoch-snark/src/api/prover.rs:78,112 -> ark_groth16::create_proof_no_zk).

CPU: the oracle's restatement is pinned on the DEFINITION - for a satisfied QAP the witness map's output is the quotient
h(x) = (a(x) b(x) - c(x)) / (x^n - 1), checked by plain polynomial multiplication (no FFT code involved).
GPU: groth16_witness_map_bw6_761 and groth16_prove_bw6_761 against that oracle, bit for bit."""
import numpy as np
import pytest
from oracle.py import ecc, ntt as ontt, groth16_prover as gp
from oracle import cpu_oracle as co

Q = ecc.Q377


def _rand(rng, n):
    return [ecc.random_scalar(rng, Q) for _ in range(n)]


def _coset_generator():
    g = 2                                            # any element outside the 2^k-torsion works as the coset offset; arkworks uses
    while pow(g, (Q - 1) // 2, Q) != Q - 1:          # F::multiplicative_generator(), the caller of the C ABI passes its own
        g += 1
    return g


def _poly_mul(a, b):
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % Q
    return r


def test_oracle_witness_map_is_the_quotient_by_the_vanishing_polynomial():
    log_n, n = 4, 16
    w, g = ontt.root_of_unity(log_n), _coset_generator()
    rng = ecc.SplitMix64(808)
    a, b = _rand(rng, n), _rand(rng, n)
    c = [x * y % Q for x, y in zip(a, b)]            # a satisfied QAP: a o b = c on the domain
    h = gp.witness_map(a, b, c, log_n, w, g)
    winv, ninv = pow(w, -1, Q), pow(n, -1, Q)
    coef = lambda v: [x * ninv % Q for x in ontt.dft(v, winv)]        # interpolation by the O(n^2) definition
    A, B, Cc = coef(a), coef(b), coef(c)
    lhs = _poly_mul(A, B)
    for i, x in enumerate(Cc):
        lhs[i] = (lhs[i] - x) % Q
    rhs = [0] * (2 * n - 1)                          # h(x) * (x^n - 1)
    for i, x in enumerate(h):
        if i + n < len(rhs):
            rhs[i + n] = (rhs[i + n] + x) % Q
        else:
            assert x == 0                            # deg h <= n - 2
        rhs[i] = (rhs[i] - x) % Q
    assert lhs == rhs
    assert h[n - 1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [6, 12, 16])
def test_witness_map_on_gpu_matches_oracle(gpu, log_n):
    n = 1 << log_n
    w, g = ontt.root_of_unity(log_n), _coset_generator()
    rng = np.random.default_rng(100 + log_n)
    def felts():
        x = rng.integers(0, 1 << 62, size=(n, 6), dtype=np.int64).astype(np.uint64)
        x[:, 5] &= np.uint64((1 << 56) - 1)          # < 2^376 < q: valid Montgomery residues
        return x
    am, bm, cm = felts(), felts(), felts()
    a, b, c = (co.from_mont(x, Q) for x in (am, bm, cm))
    if log_n <= 12:
        c = [x * y % Q for x, y in zip(a, b)]        # satisfied QAP at the small sizes, arbitrary c at 2^16
        cm = co.to_mont(c, Q)
    want = gp.witness_map(a, b, c, log_n, w, g)
    k = gp.domain_constants(log_n, w, g)
    consts = {name: co.to_mont([v], Q)[0] for name, v in k.items()}
    got = gpu.witness_map(am, bm, cm, log_n, consts)
    assert co.from_mont(got, Q) == want
    got_c = gpu.witness_map(am, bm, cm, log_n, consts, canonical=True)
    assert co.limbs_to_ints(got_c, 6) == want
    if log_n <= 12:
        assert want[n - 1] == 0


@pytest.mark.gpu
def test_prove_no_zk_on_gpu_matches_oracle(gpu):
    """a synthetic proving key (queries = k_i * P for the two r-torsion points the reference's own verifying key holds), a
    witness-like assignment (zeros and ones among full-size scalars) and the witness map's h: A, B, C equal the oracle's."""
    import torch
    from celo_bls_snark_rs_amd import synthetic as syn
    log_n, n = 12, 4096
    n_inputs, n_aux = 2, 3500
    n_assign = n_inputs + n_aux
    def pts(group, k, seed):
        return syn.device_points(group, k, seed).cpu().numpy().view(np.uint64).reshape(k, 24)
    a_query, b_query = pts("bw6_761_g1", n_assign + 1, 11), pts("bw6_761_g2", n_assign + 1, 12)
    l_query, h_query = pts("bw6_761_g1", n_aux, 13), pts("bw6_761_g1", n - 1, 14)
    alpha, beta = syn.generator_limbs("bw6_761_g1"), syn.generator_limbs("bw6_761_g2")
    asg = syn.witness_like_scalars("bw6_761_g1", n_assign, 15)
    w, g = ontt.root_of_unity(log_n), _coset_generator()
    rng = ecc.SplitMix64(16)
    a, b = _rand(rng, n), _rand(rng, n)
    c = [x * y % Q for x, y in zip(a, b)]
    k = gp.domain_constants(log_n, w, g)
    consts = {name: co.to_mont([v], Q)[0] for name, v in k.items()}
    h = gpu.witness_map(co.to_mont(a, Q), co.to_mont(b, Q), co.to_mont(c, Q), log_n, consts, canonical=True)
    h_ints = co.limbs_to_ints(h, 6)
    assert h_ints == gp.witness_map(a, b, c, log_n, w, g)
    A, B, Cc = gpu.groth16_prove(a_query, b_query, h_query, l_query, alpha, beta, asg, n_aux, h)
    wa, wb, wc = gp.prove_no_zk(a_query, b_query, h_query, l_query, alpha, beta, co.limbs_to_ints(asg, 6), n_aux, h_ints)
    assert co.jac_to_affine(A, "761") == wa and co.jac_to_affine(B, "761") == wb and co.jac_to_affine(Cc, "761") == wc
    assert wa is not None and wc is not None
