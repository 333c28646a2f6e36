"""Host-side mirror of bls-crypto's hot-path API over the C ABI (same names, argument meaning and error behaviour as
crates/bls-crypto/src/bls/{public,signature,batch}.rs) — thin orchestration; all group arithmetic runs on the GPU.

Points are affine tuples of Python ints (G1: (x, y); G2: ((x0, x1), (y0, y1))), None = identity.  This mirror takes the
message hash point H(m) explicitly (the batched hashers are `ffi.hash_to_g1_direct` / `ffi.hash_to_g1_composite`, and the
reference-named FFI symbols of include/celo_bls_snark_sys.h hash from message bytes); `batch_verify_strict` below is the
list-of-tuples form of `ffi.batch_verify` (one chained device call on flat arrays).
"""
import os
import numpy as np
from . import codec, ffi

G2_GENERATOR = (
    (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
     140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
     149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
)
SECURITY_BOUND = 128  # crates/bls-crypto/src/bls/batch.rs:20


class BLSError(Exception):
    """crates/bls-crypto/src/lib.rs:85 BLSError::VerificationFailed and friends."""


def _neg_g2(P):
    (x0, x1), (y0, y1) = P
    q = codec.Q377
    return ((x0, x1), (-y0 % q, -y1 % q))


def byte_count_from_target_batch_size(size, target_security=SECURITY_BOUND):
    """crates/bls-crypto/src/bls/batch.rs:23-28 (ark_std::log2 is ceil-log2)."""
    lg = 0 if size <= 1 else (size - 1).bit_length()
    return min((target_security + lg + 7) // 8, 253 // 8)


def public_key_batch(exponents, public_keys):
    """PublicKey::batch (public.rs:47-65): sum e_i * pk_i, None if the lengths differ."""
    if len(exponents) != len(public_keys):
        return None
    xy, inf = codec.pack_affine(public_keys, codec.Q377, ext=2)
    out = ffi.msm("bls12_377_g2", xy, inf, codec.ints_to_limbs(exponents, 4))
    return codec.jacobian_to_affine(out, codec.Q377, ext=2)


def signature_batch(exponents, signatures):
    """Signature::batch (signature.rs:70-89)."""
    if len(exponents) != len(signatures):
        return None
    xy, inf = codec.pack_affine(signatures, codec.Q377)
    out = ffi.msm("bls12_377_g1", xy, inf, codec.ints_to_limbs(exponents, 4))
    return codec.jacobian_to_affine(out, codec.Q377)


def verify_hash(public_key, message_hash, signature):
    """PublicKey::verify_sig with the hash already computed (public.rs:94-120): e(sig,-g2) * e(H(m),pk) == 1."""
    g1, i1 = codec.pack_affine([signature, message_hash], codec.Q377)
    g2, i2 = codec.pack_affine([_neg_g2(G2_GENERATOR), public_key], codec.Q377, ext=2)
    if not ffi.pairing_product_is_one(g1, i1, g2, i2):
        raise BLSError("VerificationFailed")


def batch_verify_hashes(aggregate_signature, public_keys, message_hashes):
    """Signature::batch_verify_hashes (signature.rs:125-155): one (n+1)-pair product, ONE final exponentiation."""
    if len(public_keys) != len(message_hashes):
        raise BLSError("UnevenNumKeysMessages")
    g1, i1 = codec.pack_affine([aggregate_signature] + list(message_hashes), codec.Q377)
    g2, i2 = codec.pack_affine([_neg_g2(G2_GENERATOR)] + list(public_keys), codec.Q377, ext=2)
    if not ffi.pairing_product_is_one(g1, i1, g2, i2):
        raise BLSError("VerificationFailed")


def batch_verify_strict(batches, exponents=None):
    """The loop of bls-snark-sys batch_verify_strict (crates/bls-snark-sys/src/signatures.rs:343-400) over Batch::verify
    (batch.rs:44-84), vectorised: all batches' G2 MSMs in one call, all G1 MSMs in one call, all 2-pair checks in one call.

    batches: list of (public_keys, signatures, message_hash).  exponents: optional list of per-batch exponent lists
    (injectable for tests; production draws `byte_count_from_target_batch_size` random bytes per entry from the OS RNG,
    batch.rs:51-66).  Returns the per-batch accept list (out_results of the FFI)."""
    m = len(batches)
    if m == 0:
        return []
    if exponents is None:
        exponents = []
        for pks, _, _ in batches:
            nb = byte_count_from_target_batch_size(len(pks))
            exponents.append([int.from_bytes(os.urandom(nb), "little") for _ in pks])
    offs = np.zeros(m + 1, dtype=np.uint32)
    pk_all, sig_all, e_all = [], [], []
    for i, ((pks, sigs, _), ex) in enumerate(zip(batches, exponents)):
        if len(pks) != len(sigs) or len(ex) != len(pks):
            raise BLSError("Uneven number of exponents / public keys / signatures")
        offs[i + 1] = offs[i] + len(pks)
        pk_all += list(pks)
        sig_all += list(sigs)
        e_all += list(ex)
    sc = codec.ints_to_limbs(e_all, 4)
    pk_xy, pk_inf = codec.pack_affine(pk_all, codec.Q377, ext=2)
    sg_xy, sg_inf = codec.pack_affine(sig_all, codec.Q377)
    bpk = ffi.msm_batch("bls12_377_g2", pk_xy, pk_inf, sc, offs)
    bsg = ffi.msm_batch("bls12_377_g1", sg_xy, sg_inf, sc, offs)
    ng2 = _neg_g2(G2_GENERATOR)
    g1l, g2l = [], []
    for i, (_, _, h) in enumerate(batches):
        g1l += [codec.jacobian_to_affine(bsg[i], codec.Q377), h]
        g2l += [ng2, codec.jacobian_to_affine(bpk[i], codec.Q377, ext=2)]
    g1, i1 = codec.pack_affine(g1l, codec.Q377)
    g2, i2 = codec.pack_affine(g2l, codec.Q377, ext=2)
    ok = ffi.pairing_product_is_one_batch(g1, i1, g2, i2, np.arange(0, 2 * m + 1, 2, dtype=np.uint32))
    return [bool(x) for x in ok]
