"""ORACLE (test infrastructure only).

Python restatement of the reference's DirectHasher (Blake2s CRH + Blake2Xs-style XOF) and of try-and-increment hashing
to BLS12-377 G1 with the deployed ("compat") bit logic:
  crates/bls-crypto/src/hashers/direct.rs:8-80            crh / xof / xof_digest_length_to_node_offset
  crates/bls-crypto/src/hashers/mod.rs:38-48              hash = xof(crh(..))
  crates/bls-crypto/src/hash_to_curve/mod.rs:24-28,146    hash_length, from_random_bytes
  crates/bls-crypto/src/hash_to_curve/try_and_increment.rs:87-139
Blake2s is hand-rolled (RFC 7693) because hashlib refuses the fanout = 0 / depth = 0 parameter block the XOF uses.
Pinned on crates/bls-crypto/src/hashers/direct.rs:88-96,149-172 (tests/test_oracle_golden.py).  No reference vector fixes a
DIRECT-hasher hash-to-curve output (all of hash_to_curve/mod.rs:412-513 use the composite hasher): that part is unpinned.
"""
import struct
from .ecc import Q377, E1_377, H1_377, sqrt_fp

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0],
]
M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def _compress(h, block, t, last):
    m = list(struct.unpack("<16I", block))
    v = h[:] + IV[:]
    v[12] ^= t & M32
    v[13] ^= (t >> 32) & M32
    if last:
        v[14] ^= M32

    def G(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M32; v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32; v[b] = _rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M32; v[d] = _rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32; v[b] = _rotr(v[b] ^ v[c], 7)

    for r in range(10):
        s = SIGMA[r]
        G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]])
        G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]])
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2s(data, digest_length=32, fanout=1, depth=1, leaf_length=0, node_offset=0, node_depth=0, inner_length=0,
            salt=b"", personal=b""):
    """Unkeyed BLAKE2s with an explicit parameter block (node_offset occupies 48 bits)."""
    pb = bytes([digest_length, 0, fanout, depth]) + struct.pack("<I", leaf_length) + (node_offset & ((1 << 48) - 1)).to_bytes(6, "little")
    pb += bytes([node_depth, inner_length]) + salt.ljust(8, b"\0") + personal.ljust(8, b"\0")
    h = [IV[i] ^ struct.unpack("<I", pb[4 * i:4 * i + 4])[0] for i in range(8)]
    t = 0
    blocks = [data[i:i + 64] for i in range(0, len(data), 64)] or [b""]
    for blk in blocks[:-1]:
        t += 64
        h = _compress(h, blk, t, False)
    lastb = blocks[-1]
    t += len(lastb)
    h = _compress(h, lastb.ljust(64, b"\0"), t, True)
    return struct.pack("<8I", *h)[:digest_length]


def _node_offset(i, xof_len):
    return i | ((xof_len & 0xFF) << 32) | (((xof_len >> 8) & 0xFF) << 40)


def direct_crh(domain, message, xof_digest_length):
    return blake2s(message, digest_length=32, node_offset=_node_offset(0, xof_digest_length), personal=domain)


def direct_xof(domain, hashed, xof_digest_length):
    assert len(domain) <= 8
    n = (xof_digest_length + 31) // 32
    out = b""
    for i in range(n):
        hl = xof_digest_length % 32 if (i == n - 1 and xof_digest_length % 32) else 32
        out += blake2s(hashed, digest_length=hl, fanout=0, depth=0, leaf_length=32, inner_length=32,
                       node_offset=_node_offset(i, xof_digest_length), personal=domain)
    return out


def direct_hash(domain, message, n):
    return direct_xof(domain, direct_crh(domain, message, n), n)


def hash_length(n):
    return ((n * 8 + 255) // 256) * 256 // 8


def from_random_bytes_g1(b48):
    """ark-ff from_random_bytes_with_flags::<YSignFlags> + GroupAffine::get_point_from_x for BLS12-377 G1."""
    b = bytearray(b48)
    flags = b[47] & 0xC0
    b[47] &= 0x01                       # keep bits below MODULUS_BITS = 377
    x = int.from_bytes(b, "little")
    if x >= Q377:
        return None
    if x == 0 and (flags & 0x40):
        return "zero"
    y = sqrt_fp((x * x * x + 1) % Q377, Q377)
    if y is None:
        return None
    neg = (-y) % Q377
    greatest = bool(flags & 0x80)
    return (x, y if ((y < neg) ^ greatest) else neg)


def hash_to_g1_direct(domain, message, extra_data):
    """TryAndIncrement<DirectHasher, G1>::hash_with_attempt with the `compat` feature (the deployed behaviour)."""
    nbytes = 48
    hb = hash_length(nbytes)
    for c in range(255):
        cand = bytearray(direct_hash(domain, bytes([c]) + extra_data + message, hb)[:nbytes])
        if cand[nbytes - 1] & 2:
            cand[nbytes - 1] |= 0x80
        else:
            cand[nbytes - 1] &= 0x7F
        P = from_random_bytes_g1(bytes(cand))
        if P is None:
            continue
        if P == "zero":
            continue
        S = E1_377.mul(P, H1_377)
        if S is None:
            continue
        return S, c
    raise ValueError("HashToCurveError")


# ---------------------------------------------------------------------------------------------------------------------
# generic try-and-increment over a (crh, xof) hasher: composite = Bowe-Hopwood CRH (ignores the domain) + Blake2Xs XOF
def _hasher(composite):
    if composite:
        from .composite import composite_crh
        return (lambda dom, msg, n: composite_crh(msg)), direct_xof
    return direct_crh, direct_xof


def _candidate_to_point(cand48):
    cand = bytearray(cand48)
    if cand[47] & 2:                 # `compat` bit logic: bit 1 of the last byte selects the y sign
        cand[47] |= 0x80
    else:
        cand[47] &= 0x7F
    return from_random_bytes_g1(bytes(cand))


def ark_scale_by_cofactor_jacobian(P):
    """The Jacobian representative (X, Y, Z) that ark-ec's GroupAffine::scale_by_cofactor returns for the affine G1 point P - what
    hash_composite / hash_composite_cip22 write as 144 bytes (crates/bls-snark-sys/src/signatures.rs:143,215; try_and_increment.rs:130).
    mul_bits over BitIteratorBE(COFACTOR): res = (0, 1, 0); per bit, most significant first: res.double_in_place(); if bit:
    res.add_assign_mixed(P) - doubling dbl-2009-l (a = 0), mixed addition madd-2007-bl, as in short_weierstrass_jacobian.rs of the pinned
    arkworks revision (restated from the published formulas; SURVEY.md Appendix B.6: source not on disk, no reference vector pins the bytes)."""
    p = Q377
    px, py = P
    X, Y, Z = 0, 1, 0

    def dbl(X, Y, Z):
        if Z == 0:
            return X, Y, Z
        a, b = X * X % p, Y * Y % p
        c = b * b % p
        d = 2 * ((X + b) * (X + b) - a - c) % p
        e = 3 * a % p
        f = e * e % p
        Z3 = 2 * Y * Z % p
        X3 = (f - 2 * d) % p
        Y3 = (e * (d - X3) - 8 * c) % p
        return X3, Y3, Z3

    for i in range(127, -1, -1):                 # BitIteratorBE over the two 64-bit limbs of COFACTOR
        X, Y, Z = dbl(X, Y, Z)
        if (H1_377 >> i) & 1:
            if Z == 0:
                X, Y, Z = px, py, 1
                continue
            z1z1 = Z * Z % p
            u2 = px * z1z1 % p
            s2 = py * Z % p * z1z1 % p
            if X == u2 and Y == s2:
                X, Y, Z = dbl(X, Y, Z)
                continue
            h = (u2 - X) % p
            hh = h * h % p
            i4 = 4 * hh % p
            j = h * i4 % p
            r = 2 * (s2 - Y) % p
            v = X * i4 % p
            X3 = (r * r - j - 2 * v) % p
            Y3 = (r * (v - X3) - 2 * Y * j) % p
            Z3 = ((Z + h) * (Z + h) - z1z1 - hh) % p
            X, Y, Z = X3, Y3, Z3
    return X, Y, Z


def hash_to_g1(domain, message, extra_data, composite=False, cip22=False, want_pre=False):
    """TryAndIncrement::hash_with_attempt (try_and_increment.rs:87-139) or TryAndIncrementCIP22::hash_with_attempt_cip22
    (try_and_increment_cip22.rs:81-134), `compat` feature on.  Returns (affine point, attempt); want_pre: also the curve point before
    scale_by_cofactor."""
    crh, xof = _hasher(composite)
    hb = hash_length(48)
    inner = crh(domain, message, hb) if cip22 else None
    for c in range(255):
        if cip22:
            cand = xof(domain, bytes([c]) + extra_data + inner, hb)[:48]
        else:
            cand = xof(domain, crh(domain, bytes([c]) + extra_data + message, hb), hb)[:48]
        P = _candidate_to_point(cand)
        if P is None or P == "zero":
            continue
        S = E1_377.mul(P, H1_377)
        if S is None:
            continue
        return (S, c, P) if want_pre else (S, c)
    raise ValueError("HashToCurveError")
