"""GPU (-m gpu): batched small MSMs and the Batch::verify / batch_verify_strict flow (BASELINE config 3 shape at
oracle-checkable sizes).  Reference: crates/bls-crypto/src/bls/batch.rs:44-84, crates/bls-snark-sys/src/signatures.rs:343-400."""
import numpy as np
import pytest
from oracle.py import ecc
from oracle import cpu_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _check_batch(gpu, group, kind, curve, gen, packer, sizes, bits, seed, subgroup=False):
    rng = ecc.SplitMix64(seed)
    offs = np.zeros(len(sizes) + 1, dtype=np.uint32)
    pts, sc = [], []
    for i, k in enumerate(sizes):
        offs[i + 1] = offs[i] + k
        base = curve.mul(gen, rng.next() | 1)
        inst = []
        for j in range(k):
            inst.append(curve.add(base, inst[-1]) if inst else base)   # cheap distinct points: base, 2base, ...
        if k >= 8:
            inst[3] = None
            inst[5] = inst[6]
        pts += inst
        s = [ecc.random_scalar(rng, 1 << bits) for _ in range(k)]
        if k >= 8:
            s[0] = 0; s[1] = 1; s[5] = s[6]
        sc += s
    xy, inf = packer(pts)
    s_np = H.scalars_np(sc, 4)
    got = gpu.msm_batch(group, xy, inf, s_np, offs, subgroup=subgroup)
    for i, k in enumerate(sizes):
        lo, hi = int(offs[i]), int(offs[i + 1])
        exp = co.jac_to_affine(co.msm(group, xy[lo:hi], inf[lo:hi], s_np[lo:hi], threads=2), kind) if k else None
        assert co.jac_to_affine(got[i], kind) == exp, (i, k)


def test_batch_g1_mixed_sizes_136bit(gpu):
    _check_batch(gpu, "bls12_377_g1", "g1_377", ecc.E1_377, ecc.G1_377, co.pack_g1_377, [1, 17, 256, 0, 300, 64], 136, 1)


@pytest.mark.parametrize("bits", [1, 4, 5, 6, 31, 32, 33, 64, 65, 135, 137, 252])
def test_batch_window_count_follows_the_longest_scalar(gpu, bits):
    """the batch path sizes its window count by the longest scalar present (k_scalar_or): lengths on both sides of the 5-bit
    window and 32-bit limb boundaries, one instance whose top scalar is exactly 2^bits - 1 among shorter ones"""
    _check_batch(gpu, "bls12_377_g1", "g1_377", ecc.E1_377, ecc.G1_377, co.pack_g1_377, [9, 40, 1, 130], bits, 100 + bits)
    rng = ecc.SplitMix64(900 + bits)
    pts = [ecc.E1_377.mul(ecc.G1_377, rng.next() | 1) for _ in range(20)]
    sc = [ecc.random_scalar(rng, 1 << max(1, bits - 3)) for _ in range(20)]
    sc[13] = (1 << bits) - 1
    xy, inf = co.pack_g1_377(pts)
    s_np = H.scalars_np(sc, 4)
    offs = np.array([0, 12, 20], dtype=np.uint32)
    got = gpu.msm_batch("bls12_377_g1", xy, inf, s_np, offs)
    for i in range(2):
        lo, hi = int(offs[i]), int(offs[i + 1])
        assert co.jac_to_affine(got[i], "g1_377") == co.jac_to_affine(co.msm("bls12_377_g1", xy[lo:hi], inf[lo:hi], s_np[lo:hi], threads=1), "g1_377")


def test_batch_all_zero_scalars(gpu):
    pts = [ecc.E1_377.mul(ecc.G1_377, 5 + i) for i in range(10)]
    xy, inf = co.pack_g1_377(pts)
    got = gpu.msm_batch("bls12_377_g1", xy, inf, H.scalars_np([0] * 10, 4), np.array([0, 4, 10], dtype=np.uint32))
    assert co.jac_to_affine(got[0], "g1_377") is None and co.jac_to_affine(got[1], "g1_377") is None


def test_batch_g1_full_scalars(gpu):
    _check_batch(gpu, "bls12_377_g1", "g1_377", ecc.E1_377, ecc.G1_377, co.pack_g1_377, [33, 256, 700], 252, 2)


def test_batch_g2(gpu):
    _check_batch(gpu, "bls12_377_g2", "g2_377", ecc.E2_377, ecc.G2_377, co.pack_g2_377, [5, 64, 256], 136, 3)


@pytest.mark.parametrize("bits", [63, 64, 65, 96, 97, 126, 127, 128, 136, 160, 161, 189, 190, 192, 193, 224, 225, 252])
def test_batch_g2_subgroup_points_endomorphism_split(gpu, bits):
    """msm_batch_bls12_377_g2_subgroup: the GLS split k = d0 + d1 x + d2 x^2 + d3 x^3, [x]P = psi(P) (csrc/msm.h k_gls_expand, gls.h) for
    every scalar length class - no split up to 64 bits, 2 digits to 126, 3 to 189 (Batch::verify's 136-bit exponents), 4 beyond - on both
    sides of every border of the (significant words, digits) dispatch, with identities, repeated points, zero and unit scalars inside;
    an instance too large to be expanded within the per-workgroup sort (300 points x 4 digits) takes the unsplit path.  Same sums as
    the oracle's Pippenger, instance by instance."""
    _check_batch(gpu, "bls12_377_g2", "g2_377", ecc.E2_377, ecc.G2_377, co.pack_g2_377, [1, 9, 64, 0, 256, 300 if bits > 189 else 130], bits, 500 + bits, subgroup=True)


def test_batch_verify_strict_flow(gpu):
    """Vectorised batch_verify_strict: valid batches accept, a batch with one bad signature rejects; the same exponents go
    through the oracle's MSM + pairing and must give the same accept vector."""
    from celo_bls_snark_rs_amd import bls
    rng = ecc.SplitMix64(99)
    batches, exps, expect = [], [], []
    for b in range(6):
        n = [3, 20, 64, 7, 1, 20][b]
        h = ecc.E1_377.mul(ecc.G1_377, rng.next() | 1)         # synthetic H(m) = h*G1 (hash-to-curve is out of scope)
        sks = [ecc.random_scalar(rng, ecc.R377) for _ in range(n)]
        pks = [ecc.E2_377.mul(ecc.G2_377, sk) for sk in sks]
        sigs = [ecc.E1_377.mul(h, sk) for sk in sks]
        bad = b in (2, 4)
        if bad:
            sigs[n // 2] = ecc.E1_377.mul(h, sks[n // 2] + 1)
        batches.append((pks, sigs, h))
        nb = bls.byte_count_from_target_batch_size(n)
        exps.append([int.from_bytes(bytes((rng.next() >> (8 * (k % 8))) & 0xFF for k in range(nb)), "little") for _ in range(n)])
        expect.append(not bad)
    assert bls.byte_count_from_target_batch_size(256) == 17 and bls.byte_count_from_target_batch_size(257) == 18
    got = bls.batch_verify_strict(batches, exps)
    assert got == expect
    # oracle on the same exponents
    for (pks, sigs, h), ex, want in zip(batches, exps, expect):
        xy2, i2 = co.pack_g2_377(pks)
        xy1, i1 = co.pack_g1_377(sigs)
        s = H.scalars_np(ex, 4)
        bpk = co.jac_to_affine(co.msm("bls12_377_g2", xy2, i2, s), "g2_377")
        bsg = co.jac_to_affine(co.msm("bls12_377_g1", xy1, i1, s), "g1_377")
        g1, j1 = co.pack_g1_377([bsg, h])
        g2, j2 = co.pack_g2_377([ecc.E2_377.neg(ecc.G2_377), bpk])
        assert co.pairing_product_377(g1, j1, g2, j2)[1] == want
    # production path draws its own exponents (OS RNG): verdicts must still be right
    assert bls.batch_verify_strict(batches) == expect
    # single verify + aggregate screening mirrors
    pks, sigs, h = batches[0]
    bls.verify_hash(pks[0], h, sigs[0])
    with pytest.raises(bls.BLSError):
        bls.verify_hash(pks[1], h, sigs[0])
    assert bls.public_key_batch([1, 2], pks[:1]) is None      # length mismatch -> None (public.rs:53-56)


@pytest.mark.gpu
def test_device_exponent_kernel_is_chacha20_per_signer(gpu):
    """celo_amd_draw_batch_exponents = what batch_verify_strict draws on the device (csrc/unit_batchverify.hip k_draw_exponents): signer i
    keeps the first byte_count_from_target_batch_size(128, n) bytes (crates/bls-crypto/src/bls/batch.rs:23-28) of ChaCha20 block i under
    the call's key.  Checked against a ChaCha20 block function written here and pinned on RFC 7539 section 2.3.2's test vector."""
    import struct

    def block(key_words, counter, nonce_words):
        s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & 0xFFFFFFFF] + list(nonce_words)
        w = list(s)
        rot = lambda x, n: ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF

        def qr(a, b, c, d):
            w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rot(w[d] ^ w[a], 16)
            w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rot(w[b] ^ w[c], 12)
            w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = rot(w[d] ^ w[a], 8)
            w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = rot(w[b] ^ w[c], 7)
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(w, s)])

    rfc_key = struct.unpack("<8I", bytes(range(32)))
    rfc = block(rfc_key, 1, struct.unpack("<3I", bytes.fromhex("000000090000004a00000000")))
    assert rfc[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4" and rfc[-4:].hex() == "a2503c4e"
    key = np.array([0x01234567, 0x89ABCDEF, 0xDEADBEEF, 0x0BADF00D, 7, 0, 0xFFFFFFFF, 0x5EED5EED], dtype=np.uint32)
    sizes = [1, 2, 3, 0, 256, 257, 1024, 5, 0, 65, 70000]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    got = gpu.draw_batch_exponents(key, offsets)
    assert got.shape == (sum(sizes), 4)
    raw = got.view(np.uint8).reshape(-1, 32)
    for b, n in enumerate(sizes):
        lg = 0
        while (1 << lg) < n:
            lg += 1
        nbytes = min(31, (128 + lg + 7) // 8)
        picks = range(int(offsets[b]), int(offsets[b + 1])) if n <= 300 else [int(offsets[b]), int(offsets[b]) + 1, int(offsets[b + 1]) - 1, int(offsets[b]) + n // 2]
        for at in picks:
            want = block([int(k) for k in key], at, (0, 0, 0))[:nbytes] + bytes(32 - nbytes)
            assert bytes(raw[at]) == want, (b, at)
    # all exponents distinct, none zero, and the top byte count is what the MSM's window count will see (<= 8 * nbytes bits)
    assert len({bytes(r) for r in raw}) == raw.shape[0]
