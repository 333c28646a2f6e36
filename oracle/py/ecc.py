"""ORACLE (test infrastructure only — never imported by the product path).

Pure-Python big-integer restatement of the group law, wire format and a
*textbook* pairing for BLS12-377 and BW6-761.  It is deliberately naive and
independent of both the C++ oracle (oracle/cpu) and the HIP product
(celo-bls-snark-rs_amd/csrc): group elements have a unique affine form, so any
correct implementation must agree with this one byte-for-byte on the
arkworks wire encoding.

The arithmetic itself lives in un-vendored arkworks crates (ark-ec / ark-ff
0.1.0 @ arkworks-rs/algebra#8d76d181, ark-bls12-377 / ark-bw6-761 @
arkworks-rs/curves#6ed2450b; Cargo.lock:50-292 of the reference).  The call
sites this file serves as oracle for:
  crates/bls-crypto/src/bls/public.rs:61,102   (G2 MSM, 2-pairing check)
  crates/bls-crypto/src/bls/signature.rs:85,149 (G1 MSM, (n+1)-pairing check)
  crates/epoch-snark/src/api/verifier.rs:35    (Groth16 verify, BW6-761)
Constants are pinned against the reference's golden vectors in
tests/test_oracle_golden.py.
"""

# ----------------------------------------------------------------------------
# BLS12-377 parameters
# ----------------------------------------------------------------------------
X = 0x8508C00000000001
R377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
Q377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
assert R377 == X**4 - X**2 + 1
assert Q377 == ((X - 1) ** 2 * R377) // 3 + X

G1_377 = (
    81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
)
G2_377 = (
    (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
     140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
     149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
)
NONRES2_377 = -5 % Q377  # Fq2 = Fq[u]/(u^2 + 5)
# G2 twist: y^2 = x^3 + B', B' = 1/u = -u/5  (D-type twist, xi = u)
B2_377 = (0, (-pow(5, -1, Q377)) % Q377)
assert B2_377[1] == 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906
H1_377 = (X - 1) ** 2 // 3  # G1 cofactor

# ----------------------------------------------------------------------------
# BW6-761 parameters
# ----------------------------------------------------------------------------
Q761 = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
R761 = Q377  # scalar field of BW6-761 is the base field of BLS12-377
B1_761 = Q761 - 1  # G1: y^2 = x^3 - 1
B2_761 = 4  # G2 (M-twist, coordinates in Fq): y^2 = x^3 + 4


# ----------------------------------------------------------------------------
# generic helpers
# ----------------------------------------------------------------------------
def inv(a, p):
    return pow(a, -1, p)


def sqrt_fp(a, p):
    """Tonelli-Shanks; returns None if a is a non-residue."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    c = pow(z, t, p)
    x = pow(a, (t + 1) // 2, p)
    b = pow(a, t, p)
    m = s
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2 = b2 * b2 % p
            i += 1
        e = pow(c, 1 << (m - i - 1), p)
        x = x * e % p
        c = e * e % p
        b = b * c % p
        m = i
    return x


class Fp2:
    """Fq2 = Fq[u]/(u^2 - nr) as pairs (c0, c1); static helpers on tuples."""

    def __init__(self, p, nr):
        self.p, self.nr = p, nr % p

    def add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def neg(self, a):
        return (-a[0] % self.p, -a[1] % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] + self.nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        p = self.p
        n = (a[0] * a[0] - self.nr * a[1] * a[1]) % p
        ni = inv(n, p)
        return (a[0] * ni % p, -a[1] * ni % p)

    def is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def pow(self, a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.sqr(a)
            e >>= 1
        return r

    def sqrt(self, a):
        """Any square root (complex method generalised to u^2 = nr)."""
        p = self.p
        if self.is_zero(a):
            return (0, 0)
        if a[1] % p == 0:
            s = sqrt_fp(a[0], p)
            if s is not None:
                return (s, 0)
            # a0 = nr * t^2  -> sqrt = t*u
            t = sqrt_fp(a[0] * inv(self.nr, p) % p, p)
            return None if t is None else (0, t)
        n = (a[0] * a[0] - self.nr * a[1] * a[1]) % p
        al = sqrt_fp(n, p)
        if al is None:
            return None
        i2 = inv(2, p)
        d = (a[0] + al) * i2 % p
        x0 = sqrt_fp(d, p)
        if x0 is None:
            d = (a[0] - al) * i2 % p
            x0 = sqrt_fp(d, p)
            if x0 is None:
                return None
        x1 = a[1] * inv(2 * x0 % p, p) % p
        r = (x0, x1)
        assert self.sqr(r) == (a[0] % p, a[1] % p)
        return r


F2_377 = Fp2(Q377, NONRES2_377)


class Curve:
    """Short Weierstrass y^2 = x^3 + b, a = 0, over Fp (k=1) or Fp2 (k=2).
    Points are affine tuples (x, y) or None for infinity."""

    def __init__(self, p, b, order, f2=None):
        self.p, self.b, self.n, self.f2 = p, b, order, f2

    # field dispatch -------------------------------------------------------
    def _add(self, a, b):
        return self.f2.add(a, b) if self.f2 else (a + b) % self.p

    def _sub(self, a, b):
        return self.f2.sub(a, b) if self.f2 else (a - b) % self.p

    def _mul(self, a, b):
        return self.f2.mul(a, b) if self.f2 else a * b % self.p

    def _inv(self, a):
        return self.f2.inv(a) if self.f2 else inv(a, self.p)

    def _zero(self, a):
        return self.f2.is_zero(a) if self.f2 else a % self.p == 0

    def _small(self, k, a):
        return (k * a[0] % self.p, k * a[1] % self.p) if self.f2 else k * a % self.p

    # group law --------------------------------------------------------------
    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return self._zero(self._sub(self._mul(y, y), self._add(self._mul(self._mul(x, x), x), self.b)))

    def neg(self, P):
        if P is None:
            return None
        x, y = P
        return (x, self.f2.neg(y) if self.f2 else -y % self.p)

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if self._zero(self._sub(x1, x2)):
            if self._zero(self._add(y1, y2)):
                return None
            lam = self._mul(self._small(3, self._mul(x1, x1)), self._inv(self._small(2, y1)))
        else:
            lam = self._mul(self._sub(y2, y1), self._inv(self._sub(x2, x1)))
        x3 = self._sub(self._sub(self._mul(lam, lam), x1), x2)
        y3 = self._sub(self._mul(lam, self._sub(x1, x3)), y1)
        return (x3, y3)

    # Jacobian internals for speed (a = 0) -----------------------------------
    def _jdbl(self, P):
        X1, Y1, Z1 = P
        if self._zero(Z1):
            return P
        m, s = self._mul, self._sub
        A = m(X1, X1)
        B = m(Y1, Y1)
        C = m(B, B)
        t = self._add(X1, B)
        D = self._small(2, s(s(m(t, t), A), C))
        E = self._small(3, A)
        F = m(E, E)
        X3 = s(F, self._small(2, D))
        Y3 = s(m(E, s(D, X3)), self._small(8, C))
        Z3 = self._small(2, m(Y1, Z1))
        return (X3, Y3, Z3)

    def _jadd_affine(self, P, Q):
        """P Jacobian + Q affine (not infinity)."""
        X1, Y1, Z1 = P
        if self._zero(Z1):
            one = (1, 0) if self.f2 else 1
            return (Q[0], Q[1], one)
        m, s = self._mul, self._sub
        Z1Z1 = m(Z1, Z1)
        U2 = m(Q[0], Z1Z1)
        S2 = m(m(Q[1], Z1), Z1Z1)
        H = s(U2, X1)
        rr = s(S2, Y1)
        if self._zero(H):
            if self._zero(rr):
                return self._jdbl(P)
            zero = (0, 0) if self.f2 else 0
            one = (1, 0) if self.f2 else 1
            return (one, one, zero)
        HH = m(H, H)
        HHH = m(H, HH)
        V = m(X1, HH)
        X3 = s(s(m(rr, rr), HHH), self._small(2, V))
        Y3 = s(m(rr, s(V, X3)), m(Y1, HHH))
        Z3 = m(Z1, H)
        return (X3, Y3, Z3)

    def _to_affine(self, P):
        X1, Y1, Z1 = P
        if self._zero(Z1):
            return None
        zi = self._inv(Z1)
        zi2 = self._mul(zi, zi)
        return (self._mul(X1, zi2), self._mul(self._mul(Y1, zi2), zi))

    def mul(self, P, k):
        if P is None:
            return None
        if k < 0:
            return self.mul(self.neg(P), -k)
        zero = (0, 0) if self.f2 else 0
        one = (1, 0) if self.f2 else 1
        acc = (one, one, zero)
        for bit in bin(k)[2:] if k else "":
            acc = self._jdbl(acc)
            if bit == "1":
                acc = self._jadd_affine(acc, P)
        return self._to_affine(acc)

    def msm(self, points, scalars):
        """Naive sum of scalar muls — the definition an MSM must equal."""
        acc = None
        for P, k in zip(points, scalars):
            acc = self.add(acc, self.mul(P, k))
        return acc

    def in_subgroup(self, P):
        return self.on_curve(P) and self.mul(P, self.n) is None


E1_377 = Curve(Q377, 1, R377)
E2_377 = Curve(Q377, B2_377, R377, F2_377)
E1_761 = Curve(Q761, B1_761, R761)
E2_761 = Curve(Q761, B2_761, R761)


# ----------------------------------------------------------------------------
# arkworks CanonicalSerialize wire format
# ----------------------------------------------------------------------------
def _fp_bytes(p):
    return ((p.bit_length() + 63) // 64) * 8


def ser_fp(a, p):
    return int(a % p).to_bytes(_fp_bytes(p), "little")


def _y_is_positive(curve, y):
    """arkworks flag: y > -y lexicographically (Fq2: compare c1 first, then c0)."""
    p = curve.p
    if curve.f2:
        ny = curve.f2.neg(y)
        return (y[1], y[0]) > (ny[1], ny[0])
    return y > (-y % p)


def ser_point(curve, P, compressed=True):
    p = curve.p
    nb = _fp_bytes(p) * (2 if curve.f2 else 1)
    if P is None:
        out = bytearray(nb if compressed else 2 * nb)
        out[-1] |= 0x40
        return bytes(out)
    x, y = P
    xb = ser_fp(x[0], p) + ser_fp(x[1], p) if curve.f2 else ser_fp(x, p)
    if compressed:
        out = bytearray(xb)
        if _y_is_positive(curve, y):
            out[-1] |= 0x80
        return bytes(out)
    yb = ser_fp(y[0], p) + ser_fp(y[1], p) if curve.f2 else ser_fp(y, p)
    return xb + yb


def deser_point(curve, data, compressed=True, check_subgroup=False):
    p = curve.p
    fb = _fp_bytes(p)
    nb = fb * (2 if curve.f2 else 1)
    data = bytearray(data)
    flags = data[-1] & 0xC0
    data[-1] &= 0x3F
    assert flags != 0xC0, "sign and infinity flags both set: SWFlags::from_u8 returns None (one encoding of the identity only)"
    if flags & 0x40:
        return None
    if curve.f2:
        x = (int.from_bytes(data[0:fb], "little"), int.from_bytes(data[fb:2 * fb], "little"))
        assert x[0] < p and x[1] < p
    else:
        x = int.from_bytes(data[0:fb], "little")
        assert x < p
    if compressed:
        assert len(data) == nb
        rhs = curve._add(curve._mul(curve._mul(x, x), x), curve.b)
        y = curve.f2.sqrt(rhs) if curve.f2 else sqrt_fp(rhs, p)
        if y is None:
            raise ValueError("x not on curve")
        if _y_is_positive(curve, y) != bool(flags & 0x80):
            y = curve.f2.neg(y) if curve.f2 else -y % p
    else:
        assert len(data) == 2 * nb
        if curve.f2:
            y = (int.from_bytes(data[nb:nb + fb], "little"), int.from_bytes(data[nb + fb:], "little"))
        else:
            y = int.from_bytes(data[nb:], "little")
    P = (x, y)
    if not curve.on_curve(P):
        raise ValueError("not on curve")
    if check_subgroup and not curve.in_subgroup(P):
        raise ValueError("not in subgroup")
    return P


# ----------------------------------------------------------------------------
# splitmix64 — the seeded generator every layer (py / C++ / HIP / bench) shares
# 
# ----------------------------------------------------------------------------
M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)


def random_scalar(rng, modulus):
    """Uniform in [0, modulus): draw ceil(bits/64) words, mask to bit length, reject."""
    bits = modulus.bit_length()
    words = (bits + 63) // 64
    while True:
        v = 0
        for i in range(words):
            v |= rng.next() << (64 * i)
        v &= (1 << bits) - 1
        if v < modulus:
            return v
