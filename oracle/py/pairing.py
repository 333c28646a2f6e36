"""ORACLE (test infrastructure only).

Textbook pairings in pure Python, written as differently as possible from the
tower-based C++ oracle / HIP product so that agreement is meaningful:

 * the full extension field is a *flat* polynomial ring Fq[w]/(w^k + c)
   (BLS12-377: k=12, w^12 = -5;  BW6-761: k=6, w^6 = -4), no towers;
 * lines come from affine slopes on the twist;
 * the final exponentiation is a plain square-and-multiply by (q^k-1)/r.

What can be compared with the reference's arkworks path
(PairingEngine::product_of_pairings, called at
crates/bls-crypto/src/bls/public.rs:102 and signature.rs:149): arkworks'
BLS12 final exponentiation returns the CUBE of this textbook value (SURVEY.md
Appendix B.3), and Miller-loop outputs differ only by proper-subfield factors
which the final exponentiation kills.  So  ark_pairing(P,Q) == textbook(P,Q)^3.
"""
from .ecc import Q377, R377, X, Q761, R761, F2_377, E1_377, E2_377, E1_761, E2_761, inv


class FlatExt:
    """Fq[w]/(w^k - nr) with elements as length-k lists of ints."""

    def __init__(self, p, k, nr):
        self.p, self.k, self.nr = p, k, nr % p

    def one(self):
        return [1] + [0] * (self.k - 1)

    def mul(self, a, b):
        k, p = self.k, self.p
        t = [0] * (2 * k - 1)
        for i, ai in enumerate(a):
            if ai:
                for j, bj in enumerate(b):
                    if bj:
                        t[i + j] += ai * bj
        for i in range(2 * k - 2, k - 1, -1):
            t[i - k] += t[i] * self.nr
        return [x % p for x in t[:k]]

    def pow(self, a, e):
        r = self.one()
        for bit in bin(e)[2:]:
            r = self.mul(r, r)
            if bit == "1":
                r = self.mul(r, a)
        return r

    def frob(self, a, n=1):
        """a^(q^n) via w^(q^n) = w * (nr)^((q^n-1)/k)."""
        k, p = self.k, self.p
        g = pow(self.nr, (p**n - 1) // k, p)
        out, gi = [], 1
        for i in range(k):
            out.append(a[i] * gi % p)
            gi = gi * g % p
        return out

    def inv(self, a):
        # a^(q^k - 2); slow but only used in tests
        return self.pow(a, self.p**self.k - 2)


F12_377 = FlatExt(Q377, 12, -5)
F6_761 = FlatExt(Q761, 6, -4)


# ---------------------------------------------------------------------------
# BLS12-377: ate pairing, D-type twist, untwist (x',y') -> (x' w^2, y' w^3)
# with Fq2 element a0 + a1*u embedded as a0 + a1*w^6.
# ---------------------------------------------------------------------------
def _line_377(lam, T, P):
    """l(P) = yP - lam' * xP * w + (lam' x_T' - y_T') w^3  (flat Fq12 list)."""
    f2, p = F2_377, Q377
    xP, yP = P
    c1 = f2.neg((lam[0] * xP % p, lam[1] * xP % p))
    c3 = f2.sub(f2.mul(lam, T[0]), T[1])
    out = [0] * 12
    out[0] = yP % p
    out[1], out[7] = c1
    out[3], out[9] = c3
    return out


def miller_loop_377(P, Q):
    """f_{x,Q}(P), P in G1 affine, Q in G2 affine (on the twist)."""
    if P is None or Q is None:
        return F12_377.one()
    f2, E2, F = F2_377, E2_377, F12_377
    T = Q
    f = F.one()
    for bit in bin(X)[3:]:
        lam = f2.mul((3 * f2.sqr(T[0])[0] % Q377, 3 * f2.sqr(T[0])[1] % Q377), f2.inv(f2.add(T[1], T[1])))
        f = F.mul(F.mul(f, f), _line_377(lam, T, P))
        T = E2.add(T, T)
        if bit == "1":
            lam = f2.mul(f2.sub(Q[1], T[1]), f2.inv(f2.sub(Q[0], T[0])))
            f = F.mul(f, _line_377(lam, T, P))
            T = E2.add(T, Q)
    return f


def final_exp_377(f):
    return F12_377.pow(f, (Q377**12 - 1) // R377)


def pairing_377(P, Q):
    return final_exp_377(miller_loop_377(P, Q))


def pairing_product_377(pairs):
    F = F12_377
    f = F.one()
    for P, Q in pairs:
        f = F.mul(f, miller_loop_377(P, Q))
    return final_exp_377(f)


def tower_to_flat_377(c):
    """c[i][j][k] (i: Fq12 w-index 0/1, j: Fq6 v-index 0..2, k: Fq2 u-index 0/1)
    -> flat list:  u = w^6, v = w^2  => exponent 6k + 2j + i."""
    out = [0] * 12
    for i in range(2):
        for j in range(3):
            for k in range(2):
                out[(6 * k + 2 * j + i)] = c[i][j][k]
    return out


# ---------------------------------------------------------------------------
# BW6-761: plain ate pairing with loop count (trace - 1), M-type twist,
# untwist (x',y') -> (x'/w^2, y'/w^3), w^6 = -4.  Used only for bilinearity /
# accept-bit cross-checks of the arkworks-style optimal ate in the C++ oracle.
# ---------------------------------------------------------------------------
def _trace_761():
    # BW6-761 (El Housni-Guillevic): t = x^5 - 3x^4 + 3x^3 - x + 3 + h_t * r(x), h_t = 13
    x = X
    r = (x**6 - 2 * x**5 + 2 * x**3 + x + 1) // 3
    assert r == R761
    return x**5 - 3 * x**4 + 3 * x**3 - x + 3 + 13 * r


def _line_761(lam, T, P):
    """Untwisted: X = x' w^-2, Y = y' w^-3, slope = lam' w^-1.
    l(P) = yP - lam' w^-1 xP + (lam' x' - y') w^-3; multiply through by w^3
    (w^3 lies in a proper subfield-coset killed by the final exponentiation only
    up to a constant; we therefore scale by w^3 which has order dividing
    (q^6-1)/r) :  l*w^3 = yP w^3 - lam' xP w^2 + (lam' x' - y')."""
    p = Q761
    xP, yP = P
    out = [0] * 6
    out[3] = yP % p
    out[2] = -lam * xP % p
    out[0] = (lam * T[0] - T[1]) % p
    return out


def miller_loop_761(P, Q, loop=None):
    if P is None or Q is None:
        return F6_761.one()
    E2, F, p = E2_761, F6_761, Q761
    if loop is None:
        loop = _trace_761() - 1
    T = Q
    f = F.one()
    for bit in bin(loop)[3:]:
        lam = 3 * T[0] * T[0] * inv(2 * T[1] % p, p) % p
        f = F.mul(F.mul(f, f), _line_761(lam, T, P))
        T = E2.add(T, T)
        if bit == "1":
            lam = (Q[1] - T[1]) * inv((Q[0] - T[0]) % p, p) % p
            f = F.mul(f, _line_761(lam, T, P))
            T = E2.add(T, Q)
    return f


def final_exp_761(f):
    return F6_761.pow(f, (Q761**6 - 1) // R761)


def pairing_761(P, Q):
    return final_exp_761(miller_loop_761(P, Q))
