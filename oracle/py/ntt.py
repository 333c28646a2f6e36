"""ORACLE (test infrastructure only).

The discrete Fourier transform over Fr(BW6-761) = Fq(BLS12-377) by its O(n^2) DEFINITION, and the 2-adic roots of unity
of that field.  ark-poly 0.1 (un-vendored, Cargo.lock:213-215) evaluates the same sums with a radix-2 Cooley-Tukey inside
`Radix2EvaluationDomain::fft_in_place`, which ark-groth16's witness map (called from crates/epoch-snark/src/api/prover.rs:78)
uses seven times per proof.  The C ABI under test takes the domain generator from the caller, so no arkworks constant is
restated here: `root_of_unity` derives *a* primitive 2^k-th root from the smallest quadratic non-residue.
PARITY UNPINNED against the reference (no NTT vector exists in it)."""
from .ecc import Q377, R377

TWO_ADICITY = 46
assert (Q377 - 1) % (1 << TWO_ADICITY) == 0 and ((Q377 - 1) >> TWO_ADICITY) & 1


def root_of_unity(log_n):
    g = 2
    while pow(g, (Q377 - 1) // 2, Q377) != Q377 - 1:
        g += 1
    w = pow(g, (Q377 - 1) >> TWO_ADICITY, Q377)
    return pow(w, 1 << (TWO_ADICITY - log_n), Q377)


def dft(values, omega):
    """X_j = sum_i x_i omega^(i j)"""
    n = len(values)
    pw = [pow(omega, k, Q377) for k in range(n)]
    return [sum(values[i] * pw[(i * j) % n] for i in range(n)) % Q377 for j in range(n)]


# ---- Fr(BLS12-377) (253 bits, 2-adicity 47): the field of the hash-helper proof (crates/epoch-snark/src/api/prover.rs:83-118)
TWO_ADICITY_R = 47
assert (R377 - 1) % (1 << TWO_ADICITY_R) == 0 and ((R377 - 1) >> TWO_ADICITY_R) & 1


def root_of_unity_fr377(log_n):
    g = 2
    while pow(g, (R377 - 1) // 2, R377) != R377 - 1:
        g += 1
    w = pow(g, (R377 - 1) >> TWO_ADICITY_R, R377)
    return pow(w, 1 << (TWO_ADICITY_R - log_n), R377)


def dft_mod(values, omega, p):
    """X_j = sum_i x_i omega^(i j) mod p (the definition, any prime field)"""
    n = len(values)
    pw = [pow(omega, k, p) for k in range(n)]
    return [sum(values[i] * pw[(i * j) % n] for i in range(n)) % p for j in range(n)]
