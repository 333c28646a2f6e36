"""ORACLE (test infrastructure only): the Groth16 prover's arithmetic after R1CS synthesis, restated from ark-groth16 0.1.0
(arkworks-rs/groth16#d8acb2b2, Cargo.lock:175-177; source absent from /root/reference) as the reference calls it through
`create_proof_no_zk` at crates/epoch-snark/src/api/prover.rs:78,112.

  witness_map      R1CStoQAP::witness_map (r1cs_to_qap.rs) from `domain.ifft_in_place(&mut a)` on: three inverse FFTs, three coset
                   FFTs, ab = a o b - c, division by the vanishing polynomial on the coset (the constant g^n - 1), one coset
                   inverse FFT.  The FFTs are the oracle's own C++ decimation-in-time restatement (orc_ntt_fq377).
  prove_no_zk      create_proof with r = s = 0 (prover.rs): calculate_coeff(0, query, vk_param, assignment) = query[0] + MSM + vk_param
                   for A (G1) and B (G2), C = MSM(l_query, aux) + MSM(h_query, h); MSMs by the oracle's arkworks-windowed Pippenger.
PARITY UNPINNED against the reference (it holds no proving key and no proof-generation vector; its Groth16 vector pins the VERIFIER):
pinned here by the definition - h(x) * (x^n - 1) == a(x) b(x) - c(x) as polynomials (tests/test_prover.py) - and by the group law."""
import numpy as np

from . import ecc
from .. import cpu_oracle as co

Q = ecc.Q377          # Fr(BW6-761) = Fq(BLS12-377): the epoch proof's field (prover.rs:78)
R = ecc.R377          # Fr(BLS12-377): the hash-helper proof's field (prover.rs:83-118, create_proof_no_zk::<BLSCurve, _> at :112)


def domain_constants(log_n, omega, coset, field=Q):
    n = 1 << log_n
    return {"omega": omega, "omega_inv": pow(omega, -1, field), "coset": coset, "coset_inv": pow(coset, -1, field), "size_inv": pow(n, -1, field),
            "vanishing_inv": pow((pow(coset, n, field) - 1) % field, -1, field)}


def witness_map(a, b, c, log_n, omega, coset, field=Q):
    """a, b, c: lists of n canonical ints (QAP evaluations over the domain).  Returns h as a list of n canonical ints.
    field = Q (BW6-761's scalar field, the default) or R (BLS12-377's)."""
    k = domain_constants(log_n, omega, coset, field)
    ntt = co.ntt_fq377 if field == Q else co.ntt_fr253
    def ifft(v):
        return ntt(co.to_mont(v, field), log_n, k["omega_inv"], scale=k["size_inv"])
    def coset_fft(m):
        return ntt(m, log_n, omega, coset=coset)
    A, B, C = (co.from_mont(coset_fft(ifft(v)), field) for v in (a, b, c))
    ab = [((x * y - z) * k["vanishing_inv"]) % field for x, y, z in zip(A, B, C)]
    h = ntt(co.to_mont(ab, field), log_n, k["omega_inv"], coset=k["coset_inv"], coset_after=True, scale=k["size_inv"])
    return co.from_mont(h, field)


def ark_zero_rows(bases, p, n64):
    """uint8 flags of the rows that hold arkworks' GroupAffine::zero() as coordinates: x = 0, y = 1 (ark-ec models/short_weierstrass_jacobian.rs:
    zero() = (0, 1, infinity = true)); a proving key's queries hold it for every variable absent from the respective matrix."""
    b = np.asarray(bases, dtype=np.uint64).reshape(len(bases), -1)
    half = b.shape[1] // 2
    one = np.zeros(half, dtype=np.uint64)
    one[:n64] = co.to_mont([1], p)[0]
    return (np.all(b[:, :half] == 0, axis=1) & np.all(b[:, half:] == one, axis=1)).astype(np.uint8)


def prove_no_zk(a_query, b_g2_query, h_query, l_query, alpha_g1, beta_g2, assignment, n_aux, h, threads=8):
    """queries: (k, 24) uint64 affine Montgomery limbs; alpha / beta: (24,) limbs; assignment / h: lists of canonical ints.
    Returns (A, B, C) as affine python points (or None)."""
    def msm(bases, scalars):
        k = min(len(bases), len(scalars))
        if k == 0:
            return None
        b = np.ascontiguousarray(bases[:k])
        return co.jac_to_affine(co.msm("bw6_761_g1", b, ark_zero_rows(b, ecc.Q761, 12), co.ints_to_limbs(scalars[:k], 6), threads=threads), "761")
    def pt(limbs):
        x, y = co.from_mont(np.asarray(limbs).reshape(2, 12), ecc.Q761)
        return None if (x, y) == (0, 1) else (x, y)
    E = ecc.E1_761                       # G2's group law is the same (a = 0; b is never used by it)
    aux = assignment[len(assignment) - n_aux:]
    A = E.add(E.add(pt(a_query[0]), msm(a_query[1:], assignment)), pt(alpha_g1))
    B = E.add(E.add(pt(b_g2_query[0]), msm(b_g2_query[1:], assignment)), pt(beta_g2))
    Cc = E.add(msm(l_query, aux), msm(h_query, h))
    return A, B, Cc


def prove_no_zk_bls12_377(a_query, b_g2_query, h_query, l_query, alpha_g1, beta_g2, assignment, n_aux, h, threads=8):
    """The same composition over BLS12-377 (the hash-helper proof): a / h / l queries (k, 12) uint64 G1 points, b_g2_query (k, 24) G2
    points, alpha (12,), beta (24,); assignment / h: lists of canonical ints below r.  Returns (A in G1, B in G2, C in G1) affine."""
    def msm(group, kind, bases, scalars):
        k = min(len(bases), len(scalars))
        if k == 0:
            return None
        b = np.ascontiguousarray(bases[:k])
        return co.jac_to_affine(co.msm(group, b, ark_zero_rows(b, ecc.Q377, 6), co.ints_to_limbs(scalars[:k], 4), threads=threads), kind)
    def p1(limbs):
        x, y = co.from_mont(np.asarray(limbs).reshape(2, 6), ecc.Q377)
        return None if (x, y) == (0, 1) else (x, y)
    def p2(limbs):
        v = co.from_mont(np.asarray(limbs).reshape(4, 6), ecc.Q377)
        return None if tuple(v) == (0, 0, 1, 0) else ((v[0], v[1]), (v[2], v[3]))
    E1, E2 = ecc.E1_377, ecc.E2_377
    aux = assignment[len(assignment) - n_aux:]
    A = E1.add(E1.add(p1(a_query[0]), msm("bls12_377_g1", "g1_377", a_query[1:], assignment)), p1(alpha_g1))
    B = E2.add(E2.add(p2(b_g2_query[0]), msm("bls12_377_g2", "g2_377", b_g2_query[1:], assignment)), p2(beta_g2))
    Cc = E1.add(msm("bls12_377_g1", "g1_377", l_query, aux), msm("bls12_377_g1", "g1_377", h_query, h))
    return A, B, Cc
