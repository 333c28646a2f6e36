"""ORACLE (test infrastructure only).

Python restatement of the reference's CompositeHasher: Bowe-Hopwood-Pedersen CRH over the twisted Edwards curve
ed-on-BW6-761 (a = -1, d = 79743 over Fq of BLS12-377) with generators drawn from a ChaCha20 RNG, followed by the Blake2Xs XOF:
  crates/bls-crypto/src/hashers/composite.rs:15-98      window (93 x 560), prng(), setup_crh(), crh(), xof()
  ark-crypto-primitives 0.1.0 @ fde39ab (Cargo.lock:72-74) crh::bowe_hopwood::{create_generators, evaluate}, pedersen::bytes_to_bits
  ark-ec 0.1.0 twisted_edwards_extended::{get_point_from_x, rand}; ark-ff 0.1.0 Fp384::rand (raw Montgomery limbs from the RNG)
  rand 0.7.3 / rand_chacha 0.2.2 (ChaCha20, 64-bit counter, 4-block buffer; BlockRng next_u32 / next_u64), rand_xorshift 0.2.0
Pinned on crates/bls-crypto/src/hashers/composite.rs:105-125 (CRH of the empty message and of a XorShift-seeded message).
"""
import hashlib
import struct
from .ecc import Q377, sqrt_fp

P = Q377
ED_A = P - 1
ED_D = 79743
R384_INV = pow(1 << 384, -1, P)
M32 = 0xFFFFFFFF


class XorShiftRng:
    def __init__(self, seed16):
        self.x, self.y, self.z, self.w = struct.unpack("<4I", bytes(seed16))

    def next_u32(self):
        x = self.x
        t = (x ^ (x << 11)) & M32
        self.x, self.y, self.z = self.y, self.z, self.w
        w = self.w
        self.w = (w ^ (w >> 19) ^ (t ^ (t >> 8))) & M32
        return self.w

    def gen_u8(self):
        return self.next_u32() & 0xFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & M32


def _chacha_block(key_words, counter):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & M32, (counter >> 32) & M32, 0, 0]
    w = s[:]

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & M32; w[d] = _rotl(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & M32; w[b] = _rotl(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & M32; w[d] = _rotl(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & M32; w[b] = _rotl(w[b] ^ w[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(w[i] + s[i]) & M32 for i in range(16)]


class ChaCha20Rng:
    """rand_chacha 0.2 ChaCha20Rng behind rand_core 0.5 BlockRng (64-word buffer = 4 blocks)."""

    def __init__(self, seed32):
        self.key = struct.unpack("<8I", bytes(seed32))
        self.counter = 0
        self.buf = []
        self.idx = 64

    def _generate(self):
        self.buf = []
        for _ in range(4):
            self.buf += _chacha_block(self.key, self.counter)
            self.counter += 1
        self.idx = 0

    def next_u32(self):
        if self.idx >= 64:
            self._generate()
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self):
        if self.idx < 63:
            lo, hi = self.buf[self.idx], self.buf[self.idx + 1]
            self.idx += 2
            return (hi << 32) | lo
        if self.idx >= 64:
            self._generate()
            lo, hi = self.buf[0], self.buf[1]
            self.idx = 2
            return (hi << 32) | lo
        lo = self.buf[63]
        self._generate()
        hi = self.buf[0]
        self.idx = 1
        return (hi << 32) | lo


def composite_prng():
    seed = hashlib.blake2s(b"ULTRALIGHT PRNG SEED", digest_size=32, person=b"UL_prngs").digest()
    return ChaCha20Rng(seed)


# ---- twisted Edwards arithmetic a x^2 + y^2 = 1 + d x^2 y^2 (affine, big ints)
def ed_add(p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    t = ED_D * x1 * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, P) % P
    y3 = (y1 * y2 - ED_A * x1 * x2) * pow(1 - t, -1, P) % P
    return (x3, y3)


def ed_neg(p):
    return ((-p[0]) % P, p[1])


ED_ZERO = (0, 1)


def fq_rand(rng):
    """ark-ff Fp384::rand: six next_u64 limbs taken as the MONTGOMERY representation, top 7 bits masked, rejection if >= q."""
    while True:
        limbs = [rng.next_u64() for _ in range(6)]
        limbs[5] &= (1 << 57) - 1
        raw = sum(l << (64 * i) for i, l in enumerate(limbs))
        if raw < P:
            return raw * R384_INV % P


def ed_rand(rng):
    while True:
        x = fq_rand(rng)
        greatest = bool(rng.next_u32() >> 31)
        x2 = x * x % P
        num = (ED_A * x2 - 1) % P
        den = (ED_D * x2 - 1) % P
        if den == 0:
            continue
        y2 = num * pow(den, -1, P) % P
        y = sqrt_fp(y2, P)
        if y is None:
            continue
        negy = (-y) % P
        y = y if ((y < negy) ^ greatest) else negy
        pt = (x, y)
        for _ in range(3):           # scale_by_cofactor: * 8
            pt = ed_add(pt, pt)
        return pt


WINDOW_SIZE, NUM_WINDOWS = 93, 560
_GENS = None


def generators(num_windows=NUM_WINDOWS):
    """bowe_hopwood::CRH::create_generators: one random base per window; only the first generator of each window is kept
    here (the j-th is 16^j times it) and the powers are formed lazily in crh()."""
    global _GENS
    if _GENS is None or len(_GENS) < num_windows:
        rng = composite_prng()
        _GENS = [ed_rand(rng) for _ in range(num_windows)]
    return _GENS


def composite_crh(message):
    bits = []
    for byte in message:
        for i in range(8):
            bits.append((byte >> i) & 1)
    while len(bits) % 3:
        bits.append(0)
    nseg = (len(bits) + 3 * WINDOW_SIZE - 1) // (3 * WINDOW_SIZE)
    gens = generators(max(nseg, 1))
    total = ED_ZERO
    for s in range(nseg):
        seg = bits[s * 3 * WINDOW_SIZE:(s + 1) * 3 * WINDOW_SIZE]
        g = gens[s]
        for j in range(0, len(seg), 3):
            c0, c1, c2 = seg[j], seg[j + 1], seg[j + 2]
            enc = g
            if c0:
                enc = ed_add(enc, g)
            if c1:
                enc = ed_add(enc, ed_add(g, g))
            if c2:
                enc = ed_neg(enc)
            total = ed_add(total, enc)
            for _ in range(4):
                g = ed_add(g, g)
    return total[0].to_bytes(48, "little")
