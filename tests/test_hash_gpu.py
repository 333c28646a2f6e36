"""GPU (-m gpu): batched hash-to-G1 over the direct hasher through the C ABI (include/celo_bls_amd.h:
hash_to_g1_direct_bls12_377) vs the oracle's restatement (oracle/py/hashing.py: hash_to_g1_direct) - SURVEY.md section 8f
row f1.

What it replaces: TryAndIncrement<DirectHasher, G1>::hash_with_attempt (crates/bls-crypto/src/hash_to_curve/
try_and_increment.rs:87-139, hashers/direct.rs:23-80), one call per message in Signature::batch_verify
(bls/signature.rs:111-114).  The reference holds vectors for Blake2s / the XOF (hashers/direct.rs:88-172, pinned in
tests/test_oracle_golden.py and tests/test_seam_a.py) and for the composite hash-to-curve, none for a direct-hasher point:
the oracle is the checker here.  Byte and integer work: bit-exact, including the attempt counter."""
import numpy as np
import pytest
import torch  # before the library: both must share one HIP runtime
from oracle.py import ecc, hashing as hs
from oracle import cpu_oracle as co

pytestmark = pytest.mark.gpu
SIG, POP = b"ULforxof", b"ULforpop"


def _want(dom, msgs, extras):
    pts, att = [], []
    for m, e in zip(msgs, extras):
        P, c = hs.hash_to_g1(dom, m, e, composite=False)
        pts.append(P)
        att.append(c)
    return co.pack_g1_377(pts)[0], att


def test_message_lengths_around_the_block_size(gpu):
    """counter || extra || message lengths 1 ... 200 cross the 64-byte Blake2s block boundary at every residue (incl. the
    exact multiples, where the final block is full), with and without extra data"""
    rng = np.random.default_rng(5)
    msgs = [bytes(rng.integers(0, 256, size=l, dtype=np.uint8)) for l in list(range(0, 70)) + [126, 127, 128, 129, 191, 192, 200]]
    extras = [bytes(rng.integers(0, 256, size=(i * 7) % 40, dtype=np.uint8)) for i in range(len(msgs))]
    for dom in (SIG, POP):
        xy, att = gpu.hash_to_g1_direct(dom, msgs, extras)
        wxy, watt = _want(dom, msgs, extras)
        assert att.tolist() == watt and np.array_equal(xy, wxy)
    xy, att = gpu.hash_to_g1_direct(SIG, msgs)                         # extra_off == NULL
    wxy, watt = _want(SIG, msgs, [b""] * len(msgs))
    assert att.tolist() == watt and np.array_equal(xy, wxy)
    assert max(watt) >= 2                                             # several counters were needed somewhere


def test_points_are_in_the_subgroup_and_distinct(gpu):
    msgs = [b"epoch %d" % i for i in range(300)]
    xy, att = gpu.hash_to_g1_direct(SIG, msgs, [b"\x01\x02"] * 300)
    assert (att < 255).all()
    vals = co.from_mont(xy.reshape(-1, 6), ecc.Q377)
    pts = [(vals[2 * i], vals[2 * i + 1]) for i in range(300)]
    assert len(set(pts)) == 300
    for P in pts[:16]:
        assert ecc.E1_377.on_curve(P) and ecc.E1_377.in_subgroup(P)
    wxy, watt = _want(SIG, msgs[:24], [b"\x01\x02"] * 24)
    assert np.array_equal(xy[:24], wxy) and att[:24].tolist() == watt


def test_empty_batch(gpu):
    xy, att = gpu.hash_to_g1_direct(SIG, [])
    assert xy.shape == (0, 12) and att.shape == (0,)
