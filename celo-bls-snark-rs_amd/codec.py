"""Host plumbing: arkworks in-memory / wire formats.

Product code (no oracle imports).  Pure-Python big-int conversions between
 * python ints / affine tuples,
 * arkworks Montgomery limb buffers (numpy uint64; R = 2^384 for BLS12-377 Fq, 2^768 for BW6-761 Fq),
 * CanonicalSerialize bytes (compressed points: x little-endian, flags 0x80 = y lexicographically
   largest, 0x40 = infinity, in the top bits of the last byte) — mirrors
   crates/bls-crypto/src/bls/public.rs:123-149 / signature.rs:31-57 (de)serialisation.
"""
import numpy as np

Q377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
R377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
Q761 = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
R761 = Q377

_FIELDS = {Q377: (6, 1 << 384), Q761: (12, 1 << 768)}


def ints_to_limbs(vals, nlimbs):
    buf = b"".join(int(v).to_bytes(8 * nlimbs, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), nlimbs).copy()


def limbs_to_ints(arr, nlimbs):
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, nlimbs)
    raw = a.tobytes()
    step = 8 * nlimbs
    return [int.from_bytes(raw[i * step:(i + 1) * step], "little") for i in range(a.shape[0])]


def to_mont(vals, p):
    n, R = _FIELDS[p]
    return ints_to_limbs([(v * R) % p for v in vals], n)


def from_mont(arr, p):
    n, R = _FIELDS[p]
    Rinv = pow(R, -1, p)
    return [(v * Rinv) % p for v in limbs_to_ints(arr, n)]


def pack_affine(points, p, ext=1):
    """points: list of affine tuples (coordinates ints, or pairs for ext=2) or None.
    Returns (xy limbs [n, 2*ext*N], inf bytes [n])."""
    flat, inf = [], []
    for P in points:
        if P is None:
            flat += [0] * (2 * ext)
            inf.append(1)
        else:
            x, y = P
            flat += ([x, y] if ext == 1 else [x[0], x[1], y[0], y[1]])
            inf.append(0)
    n = _FIELDS[p][0]
    return to_mont(flat, p).reshape(len(points), 2 * ext * n), np.array(inf, dtype=np.uint8)


def jacobian_to_affine(arr, p, ext=1):
    """One Jacobian point (X,Y,Z Montgomery limbs) -> affine tuple or None."""
    n = _FIELDS[p][0]
    v = from_mont(np.asarray(arr).reshape(3 * ext, n), p)
    if ext == 1:
        X, Y, Z = v
        if Z == 0:
            return None
        zi = pow(Z, -1, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)
    nr = -5 % p  # Fq2 = Fq[u]/(u^2 + 5)

    def mul(a, b):
        return ((a[0] * b[0] + nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    X, Y, Z = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
    if Z == (0, 0):
        return None
    nrm = (Z[0] * Z[0] - nr * Z[1] * Z[1]) % p
    ni = pow(nrm, -1, p)
    zi = (Z[0] * ni % p, -Z[1] * ni % p)
    zi2 = mul(zi, zi)
    return (mul(X, zi2), mul(Y, mul(zi2, zi)))


def decompress_bw6_761(data, g2=False):
    """96-byte arkworks compressed BW6-761 point -> affine tuple.  G1: y^2 = x^3 - 1; G2 (M-twist, coordinates in Fq):
    y^2 = x^3 + 4.  q = 3 (mod 4), so the square root is one exponentiation.  (VerifyingKey / Proof points of
    crates/bls-snark-sys/src/snark/mod.rs:23-45 arrive in this form.)"""
    p = Q761
    b = bytearray(data)
    flags = b[-1] & 0xC0
    b[-1] &= 0x3F
    if flags & 0x40:
        return None
    x = int.from_bytes(b, "little")
    if x >= p:
        raise ValueError("non-canonical x")
    rhs = (x * x * x + (4 if g2 else p - 1)) % p
    y = pow(rhs, (p + 1) // 4, p)
    if y * y % p != rhs:
        raise ValueError("x not on curve")
    if (y > (p - 1) // 2) != bool(flags & 0x80):
        y = p - y
    return (x, y)
