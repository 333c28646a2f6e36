"""Groth16 prover row: the witness map and create_proof_no_zk's group arithmetic
(crates/epoch-snark/src/api/prover.rs:78,112 -> ark_groth16::create_proof_no_zk).

CPU: the oracle's restatement is pinned on the DEFINITION - for a satisfied QAP the witness map's output is the quotient
h(x) = (a(x) b(x) - c(x)) / (x^n - 1), checked by plain polynomial multiplication (no FFT code involved).
GPU: groth16_witness_map_bw6_761 and groth16_prove_bw6_761 against that oracle, bit for bit - and the same three steps over
BLS12-377 (the hash-helper proof, prover.rs:83-118: Fr(BLS12-377) transforms, G1 / G2 MSMs): ntt_bls12_377_fr,
groth16_witness_map_bls12_377, groth16_prove_bls12_377."""
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
    # a real proving key holds the point at infinity for every variable absent from A / B / the auxiliary part, as arkworks'
    # GroupAffine::zero() = (0, 1, infinity): rows x = 0, y = 1 are the identity, whatever their scalar (here full-size ones) - in
    # every query, in query[0] and in a key element
    one = co.to_mont([1], ecc.Q761)[0]
    zero_row = np.concatenate([np.zeros(12, dtype=np.uint64), one])
    a2, b2, l2, h2 = a_query.copy(), b_query.copy(), l_query.copy(), h_query.copy()
    asg2 = asg.copy()
    big = co.ints_to_limbs([ecc.Q377 - 5, ecc.Q377 // 3, 12345678901234567890123], 6)
    for rows, q in (((0, 6, 1000), a2), ((8, 1001), b2), ((4, 3499), l2), ((11, 4000), h2)):
        for r_ in rows:
            q[r_] = zero_row
    asg2[5], asg2[7], asg2[n_inputs + 4] = big[0], big[1], big[2]
    A, B, Cc = gpu.groth16_prove(a2, b2, h2, l2, alpha, zero_row, asg2, n_aux, h)
    wa2, wb2, wc2 = gp.prove_no_zk(a2, b2, h2, l2, alpha, zero_row, co.limbs_to_ints(asg2, 6), n_aux, h_ints)
    assert co.jac_to_affine(A, "761") == wa2 and co.jac_to_affine(B, "761") == wb2 and co.jac_to_affine(Cc, "761") == wc2
    assert (wa2, wb2, wc2) != (wa, wb, wc)
    # the same proofs against a LOADED key (groth16_load_key_bw6_761 + groth16_prove_with_key: the queries' fixed-base tables built once),
    # two assignments on one key, identity rows included
    for wb_ in (0, 17):
        key = gpu.ProvingKey("bw6_761", a2, b2, h2, l2, alpha, zero_row, window_bits=wb_)
        A, B, Cc = key.prove(asg2, n_aux, h)
        assert co.jac_to_affine(A, "761") == wa2 and co.jac_to_affine(B, "761") == wb2 and co.jac_to_affine(Cc, "761") == wc2
        A, B, Cc = key.prove(asg, n_aux, h[: n - 50])
        wa3, wb3, wc3 = gp.prove_no_zk(a2, b2, h2, l2, alpha, zero_row, co.limbs_to_ints(asg, 6), n_aux, h_ints[: n - 50])
        assert co.jac_to_affine(A, "761") == wa3 and co.jac_to_affine(B, "761") == wb3 and co.jac_to_affine(Cc, "761") == wc3
        key.release()


# ------------------------------------------------------------------------------------------------ the hash-helper proof: BLS12-377
R = ecc.R377


def _coset_generator_r():
    g = 2
    while pow(g, (R - 1) // 2, R) != R - 1:
        g += 1
    return g


def test_oracle_fr377_ntt_is_the_definition_and_its_witness_map_the_quotient():
    """orc_ntt_fr253 against the O(n^2) definition (all four transforms), then the oracle's witness map over Fr(BLS12-377) against
    h(x) (x^n - 1) = a(x) b(x) - c(x) by plain polynomial multiplication."""
    log_n, n = 4, 16
    w, g = ontt.root_of_unity_fr377(log_n), _coset_generator_r()
    assert pow(w, n, R) == 1 and pow(w, n // 2, R) == R - 1
    rng = ecc.SplitMix64(909)
    x = [ecc.random_scalar(rng, R) for _ in range(n)]
    xm = co.to_mont(x, R)
    assert co.from_mont(co.ntt_fr253(xm, log_n, w), R) == ontt.dft_mod(x, w, R)
    winv, ninv, ginv = pow(w, -1, R), pow(n, -1, R), pow(g, -1, R)
    assert co.from_mont(co.ntt_fr253(xm, log_n, winv, scale=ninv), R) == [v * ninv % R for v in ontt.dft_mod(x, winv, R)]
    assert co.from_mont(co.ntt_fr253(xm, log_n, w, coset=g), R) == ontt.dft_mod([v * pow(g, i, R) % R for i, v in enumerate(x)], w, R)
    want = [v * ninv % R * pow(ginv, i, R) % R for i, v in enumerate(ontt.dft_mod(x, winv, R))]
    assert co.from_mont(co.ntt_fr253(xm, log_n, winv, coset=ginv, coset_after=True, scale=ninv), R) == want
    a, b = x, [ecc.random_scalar(rng, R) for _ in range(n)]
    c = [u * v % R for u, v in zip(a, b)]
    h = gp.witness_map(a, b, c, log_n, w, g, field=R)
    coef = lambda v: [t * ninv % R for t in ontt.dft_mod(v, winv, R)]
    A, B, Cc = coef(a), coef(b), coef(c)
    lhs = [0] * (2 * n - 1)
    for i, u in enumerate(A):
        for j, v in enumerate(B):
            lhs[i + j] = (lhs[i + j] + u * v) % R
    for i, u in enumerate(Cc):
        lhs[i] = (lhs[i] - u) % R
    rhs = [0] * (2 * n - 1)
    for i, u in enumerate(h):
        if i + n < len(rhs):
            rhs[i + n] = (rhs[i + n] + u) % R
        else:
            assert u == 0
        rhs[i] = (rhs[i] - u) % R
    assert lhs == rhs and h[n - 1] == 0


def _felts_r(rng, n):
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    x[:, 3] &= np.uint64((1 << 60) - 1)              # < 2^252 < r: valid Montgomery residues
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 10, 12, 15, 16])
def test_fr377_ntt_on_gpu_matches_oracle(gpu, log_n):
    """ntt_bls12_377_fr: fft, coset_fft, ifft and coset_ifft bit for bit against the oracle's decimation-in-time restatement over the
    same field (every launch mix of the tiled passes: 8+2, 8+4, 8+6+1, 8+8 levels)."""
    n = 1 << log_n
    w = ontt.root_of_unity_fr377(log_n)
    winv, ninv, g = pow(w, -1, R), pow(n, -1, R), 22
    ginv = pow(g, -1, R)
    x = _felts_r(np.random.default_rng(300 + log_n), n)
    m1 = lambda v: co.to_mont([v % R], R)[0]
    for kw in (dict(), dict(coset=g), dict(omega=winv, scale=ninv), dict(omega=winv, coset=ginv, coset_after=True, scale=ninv)):
        om = kw.get("omega", w)
        got = gpu.ntt_fr377(x, log_n, m1(om), None if "coset" not in kw else m1(kw["coset"]), kw.get("coset_after", False),
                            None if "scale" not in kw else m1(kw["scale"]))
        want = co.ntt_fr253(x, log_n, om, kw.get("coset"), kw.get("coset_after", False), kw.get("scale"))
        assert np.array_equal(got, want), kw


@pytest.mark.gpu
def test_fr377_ntt_two_to_20_round_trip_device_resident(gpu):
    import torch
    log_n = 20
    n = 1 << log_n
    w = ontt.root_of_unity_fr377(log_n)
    m1 = lambda v: co.to_mont([v % R], R)[0]
    x = _felts_r(np.random.default_rng(320), n)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    gpu.ntt_fr377_dev(d.data_ptr(), log_n, m1(w))
    mid = d.cpu().numpy().view(np.uint64)
    assert not np.array_equal(mid, x)
    # X_0 = sum of the inputs (the definition's first row) pins the forward transform at full size
    assert co.from_mont(mid[:1], R)[0] == sum(co.from_mont(x, R)) % R
    gpu.ntt_fr377_dev(d.data_ptr(), log_n, m1(pow(w, -1, R)), None, False, m1(pow(n, -1, R)))
    assert np.array_equal(d.cpu().numpy().view(np.uint64), x)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [6, 12, 16])
def test_fr377_witness_map_on_gpu_matches_oracle(gpu, log_n):
    n = 1 << log_n
    w, g = ontt.root_of_unity_fr377(log_n), _coset_generator_r()
    rng = np.random.default_rng(400 + log_n)
    am, bm, cm = _felts_r(rng, n), _felts_r(rng, n), _felts_r(rng, n)
    a, b, c = (co.from_mont(x, R) for x in (am, bm, cm))
    if log_n <= 12:
        c = [x * y % R for x, y in zip(a, b)]
        cm = co.to_mont(c, R)
    want = gp.witness_map(a, b, c, log_n, w, g, field=R)
    k = gp.domain_constants(log_n, w, g, field=R)
    consts = {name: co.to_mont([v], R)[0] for name, v in k.items()}
    got = gpu.witness_map_fr377(am, bm, cm, log_n, consts)
    assert co.from_mont(got, R) == want
    got_c = gpu.witness_map_fr377(am, bm, cm, log_n, consts, canonical=True)
    assert co.limbs_to_ints(got_c, 4) == want
    if log_n <= 12:
        assert want[n - 1] == 0


@pytest.mark.gpu
def test_prove_no_zk_bls12_377_on_gpu_matches_oracle(gpu):
    """create_proof_no_zk::<BLSCurve, _> after synthesis (prover.rs:112): a synthetic proving key over BLS12-377 (G1 / G2 multiples of
    the generators), a witness-like assignment, the GPU witness map's h: A, B, C equal the oracle's composition."""
    from celo_bls_snark_rs_amd import synthetic as syn
    log_n, n = 12, 4096
    n_inputs, n_aux = 2, 3500
    n_assign = n_inputs + n_aux
    def pts(group, k, seed):
        A = 12 if group.endswith("g1") else 24
        return syn.device_points(group, k, seed).cpu().numpy().view(np.uint64).reshape(k, A)
    a_query, b_query = pts("bls12_377_g1", n_assign + 1, 21), pts("bls12_377_g2", n_assign + 1, 22)
    l_query, h_query = pts("bls12_377_g1", n_aux, 23), pts("bls12_377_g1", n - 1, 24)
    alpha, beta = pts("bls12_377_g1", 1, 25)[0], pts("bls12_377_g2", 1, 26)[0]
    asg = syn.witness_like_scalars("bls12_377_g1", n_assign, 27)
    w, g = ontt.root_of_unity_fr377(log_n), _coset_generator_r()
    rng = ecc.SplitMix64(28)
    a, b = [ecc.random_scalar(rng, R) for _ in range(n)], [ecc.random_scalar(rng, R) for _ in range(n)]
    c = [x * y % R for x, y in zip(a, b)]
    k = gp.domain_constants(log_n, w, g, field=R)
    consts = {name: co.to_mont([v], R)[0] for name, v in k.items()}
    h = gpu.witness_map_fr377(co.to_mont(a, R), co.to_mont(b, R), co.to_mont(c, R), log_n, consts, canonical=True)
    h_ints = co.limbs_to_ints(h, 4)
    assert h_ints == gp.witness_map(a, b, c, log_n, w, g, field=R)
    A, B, Cc = gpu.groth16_prove_bls12_377(a_query, b_query, h_query, l_query, alpha, beta, asg, n_aux, h)
    wa, wb, wc = gp.prove_no_zk_bls12_377(a_query, b_query, h_query, l_query, alpha, beta, co.limbs_to_ints(asg, 4), n_aux, h_ints)
    assert co.jac_to_affine(A, "g1_377") == wa and co.jac_to_affine(B, "g2_377") == wb and co.jac_to_affine(Cc, "g1_377") == wc
    assert wa is not None and wb is not None and wc is not None
    # the shorter of bases and scalars decides each MSM's length (VariableBaseMSM::multi_scalar_mul): drop the last 100 h coefficients
    A2, B2, C2 = gpu.groth16_prove_bls12_377(a_query, b_query, h_query, l_query, alpha, beta, asg, n_aux, h[: n - 101])
    _, _, wc2 = gp.prove_no_zk_bls12_377(a_query, b_query, h_query, l_query, alpha, beta, co.limbs_to_ints(asg, 4), n_aux, h_ints[: n - 101])
    assert co.jac_to_affine(C2, "g1_377") == wc2 and co.jac_to_affine(A2, "g1_377") == wa


@pytest.mark.gpu
def test_prove_bls12_377_identity_rows_under_the_glv_split(gpu):
    """proving-key queries hold the point at infinity (arkworks' (0, 1, infinity)) for variables absent from a matrix.  Queries
    of 2^14 terms and more take the GLV / psi^2 split of the subgroup entry points: identity rows (x = 0, y = 1) must contribute nothing
    there either, with full-size scalars against them - in every query, in query[0] and in beta."""
    from celo_bls_snark_rs_amd import synthetic as syn
    n_inputs, n_aux, n_h = 2, 17000, (1 << 15) - 1
    n_assign = n_inputs + n_aux
    def pts(group, k, seed):
        A = 12 if group.endswith("g1") else 24
        return syn.device_points(group, k, seed).cpu().numpy().view(np.uint64).reshape(k, A)
    a_query, b_query = pts("bls12_377_g1", n_assign + 1, 31), pts("bls12_377_g2", n_assign + 1, 32)
    l_query, h_query = pts("bls12_377_g1", n_aux, 33), pts("bls12_377_g1", n_h, 34)
    alpha = pts("bls12_377_g1", 1, 35)[0]
    asg = syn.uniform_scalars("bls12_377_g1", n_assign, 37)
    h = syn.uniform_scalars("bls12_377_g1", n_h, 38)
    one = co.to_mont([1], ecc.Q377)[0]
    z1 = np.concatenate([np.zeros(6, dtype=np.uint64), one])
    z2 = np.concatenate([np.zeros(12, dtype=np.uint64), one, np.zeros(6, dtype=np.uint64)])
    for r_ in (0, 9, 16000):
        a_query[r_] = z1
    for r_ in (3, 16999):
        b_query[r_] = z2
    l_query[5] = z1; l_query[16998] = z1
    h_query[0] = z1; h_query[30000] = z1
    A, B, Cc = gpu.groth16_prove_bls12_377(a_query, b_query, h_query, l_query, alpha, z2, asg, n_aux, h)
    wa, wb, wc = gp.prove_no_zk_bls12_377(a_query, b_query, h_query, l_query, alpha, z2, co.limbs_to_ints(asg, 4), n_aux, co.limbs_to_ints(h, 4))
    assert co.jac_to_affine(A, "g1_377") == wa and co.jac_to_affine(B, "g2_377") == wb and co.jac_to_affine(Cc, "g1_377") == wc
    assert wa is not None and wb is not None and wc is not None
    # ... and against the loaded key (groth16_load_key_bls12_377): G1 tables for a / l / h, a G2 table for b
    key = gpu.ProvingKey("bls12_377", a_query, b_query, h_query, l_query, alpha, z2)
    A, B, Cc = key.prove(asg, n_aux, h)
    assert co.jac_to_affine(A, "g1_377") == wa and co.jac_to_affine(B, "g2_377") == wb and co.jac_to_affine(Cc, "g1_377") == wc
    key.release()
