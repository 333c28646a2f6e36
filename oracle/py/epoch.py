"""ORACLE (test infrastructure only).

Python restatement of the epoch-block encoding and Groth16 public-input packing
needed to replay the reference's only end-to-end pairing known-answer vector
(crates/bls-snark-sys/src/snark/mod.rs:52-119).  Follows
  crates/epoch-snark/src/encoding.rs:23-83           encode_public_key / encode_u8/u16/u32
  crates/epoch-snark/src/epoch_block.rs:106-236      encode_to_bits_cip22, hash_first_last_epoch_block, hash_to_bits
  crates/bls-gadgets/src/utils.rs:2-56               bit/byte order helpers
  crates/epoch-snark/src/gadgets/mod.rs:75-83        pack (CAPACITY = 376-bit big-endian chunks)
  crates/epoch-snark/src/api/verifier.rs:23-40       verify
"""
import hashlib
from .ecc import Q377, Q761, E1_761, E2_761, E2_377, G2_377, deser_point

OUT_DOMAIN = b"ULforout"
ENTROPY_BYTES = 16
HALF = (Q377 - 1) // 2


def bytes_le_to_bits_be(b, take):
    bits = []
    for byte in b:
        for i in range(8):
            bits.append((byte >> i) & 1)
    return bits[:take][::-1]


def bytes_le_to_bits_le(b, take):
    return bytes_le_to_bits_be(b, take)[::-1]


def bits_be_to_bytes_le(bits):
    rev = bits[::-1]
    out = bytearray()
    for i in range(0, len(rev), 8):
        chunk = rev[i:i + 8]
        out.append(sum(bit << k for k, bit in enumerate(chunk)) & 0xFF)
    return bytes(out)


def encode_uint(num, nbytes):
    return bytes_le_to_bits_le(int(num).to_bytes(nbytes, "little"), 8 * nbytes)


def encode_public_key(pk):
    """pk: affine G2 point ((x0,x1),(y0,y1)) of BLS12-377, or None for the identity: the reference reads x and y of `into_affine()`
    (crates/epoch-snark/src/encoding.rs:23-47), which for the identity is arkworks' GroupAffine::zero() = (0, 1, infinity)."""
    (x0, x1), (y0, y1) = pk if pk is not None else ((0, 0), (1, 0))
    over_half = y1 > HALF or (y1 == 0 and y0 > HALF)
    bits = bytes_le_to_bits_be(x0.to_bytes(48, "little"), 377)
    bits += bytes_le_to_bits_be(x1.to_bytes(48, "little"), 377)
    bits.append(1 if over_half else 0)
    return bits


def encode_entropy(entropy):
    if entropy is None:
        entropy = bytes(ENTROPY_BYTES * 8)  # sic: the reference allocates 128 zero BYTES then takes 128 bits
    return bytes_le_to_bits_le(entropy, ENTROPY_BYTES * 8)


class EpochBlock:
    def __init__(self, index, round_, epoch_entropy, parent_entropy, maximum_non_signers, maximum_validators, pubkeys):
        self.index, self.round = index, round_
        self.epoch_entropy, self.parent_entropy = epoch_entropy, parent_entropy
        self.maximum_non_signers, self.maximum_validators = maximum_non_signers, maximum_validators
        self.pubkeys = pubkeys

    def encode_to_bits(self):
        bits = encode_uint(self.index, 2) + encode_uint(self.maximum_non_signers, 4)
        for pk in self.pubkeys:
            bits += encode_public_key(pk)
        return bits

    def encode_to_bits_cip22(self, first):
        bits = encode_uint(self.index, 2)
        bits += encode_entropy(self.parent_entropy if first else self.epoch_entropy)
        bits += encode_uint(self.maximum_non_signers, 4)
        for pk in self.pubkeys:
            bits += encode_public_key(pk)
        for _ in range(max(0, self.maximum_validators - len(self.pubkeys))):
            bits += encode_public_key(G2_377)
        return bits

    def encode_first_epoch_to_bytes_cip22(self):
        return bits_be_to_bytes_le(self.encode_to_bits_cip22(True))

    def encode_to_bytes(self):
        return bits_be_to_bytes_le(self.encode_to_bits())

    def encode_last_epoch_to_bytes_with_aggregated_pk_cip22(self):
        bits = self.encode_to_bits_cip22(False)
        agg = None
        for pk in self.pubkeys:
            agg = E2_377.add(agg, pk)
        bits += encode_public_key(agg)
        return bits_be_to_bytes_le(bits)


def hash_to_bits(data):
    h = hashlib.blake2s(data, digest_size=32, person=OUT_DOMAIN).digest()
    return bytes_le_to_bits_le(h, 256)


def hash_first_last_epoch_block(first, last):
    return hash_to_bits(first.encode_first_epoch_to_bytes_cip22()) + hash_to_bits(
        last.encode_last_epoch_to_bytes_with_aggregated_pk_cip22())


def pack(bits, capacity=376):
    out = []
    for i in range(0, len(bits), capacity):
        v = 0
        for b in bits[i:i + capacity]:
            v = (v << 1) | b
        out.append(v)
    return out


def parse_vk(data):
    """arkworks Groth16 VerifyingKey<BW6_761>, compressed."""
    a = deser_point(E1_761, data[0:96])
    b = deser_point(E2_761, data[96:192])
    g = deser_point(E2_761, data[192:288])
    d = deser_point(E2_761, data[288:384])
    n = int.from_bytes(data[384:392], "little")
    abc = [deser_point(E1_761, data[392 + 96 * i:392 + 96 * (i + 1)]) for i in range(n)]
    assert len(data) == 392 + 96 * n
    return {"alpha_g1": a, "beta_g2": b, "gamma_g2": g, "delta_g2": d, "gamma_abc_g1": abc}


def parse_proof(data):
    assert len(data) == 288
    return {"a": deser_point(E1_761, data[0:96]), "b": deser_point(E2_761, data[96:192]),
            "c": deser_point(E1_761, data[192:288])}


def groth16_pairs(vk, proof, inputs):
    """The 4 (G1, G2) pairs whose pairing product is 1 iff ark_groth16::verify_proof accepts:
    e(A,B) * e(acc,-gamma) * e(C,-delta) * e(-alpha,beta) == 1   (SURVEY.md Appendix B.5)."""
    acc = vk["gamma_abc_g1"][0]
    assert len(inputs) + 1 == len(vk["gamma_abc_g1"])
    for s, P in zip(inputs, vk["gamma_abc_g1"][1:]):
        acc = E1_761.add(acc, E1_761.mul(P, s))
    return [(proof["a"], proof["b"]), (acc, E2_761.neg(vk["gamma_g2"])), (proof["c"], E2_761.neg(vk["delta_g2"])),
            (E1_761.neg(vk["alpha_g1"]), vk["beta_g2"])]


def encode_inner_to_bytes_cip22(block):
    """EpochBlock::encode_inner_to_bytes_cip22 (crates/epoch-snark/src/epoch_block.rs:150-169, 207-214): (inner, extra_data)."""
    extra = encode_uint(block.index, 2) + encode_uint(block.round, 1) + encode_uint(block.maximum_non_signers, 4)
    bits = encode_entropy(block.epoch_entropy) + encode_entropy(block.parent_entropy)
    for pk in block.pubkeys:
        bits += encode_public_key(pk)
    for _ in range(max(0, block.maximum_validators - len(block.pubkeys))):
        bits += encode_public_key(G2_377)
    return bits_be_to_bytes_le(bits), bits_be_to_bytes_le(extra)
