"""Seam A (include/celo_bls_snark_sys.h): the bls-snark-sys symbols exported so far.  The wire-format half is host code and
runs without a GPU; it is pinned on the reference's own golden points (crates/bls-crypto/src/hash_to_curve/mod.rs:412-513:
every expected hash is a compressed G1 / G2 point) exactly like the FFI round-trip tests of
crates/bls-snark-sys/src/utils.rs:105-132 and serialization.rs."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from oracle.py import ecc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sys_lib():
    from celo_bls_snark_rs_amd import ffi
    lib = C.CDLL(ffi.LIB_PATH)
    for name in ("deserialize_public_key", "deserialize_signature", "deserialize_private_key", "serialize_public_key",
                 "serialize_signature", "serialize_private_key", "serialize_public_key_uncompressed",
                 "serialize_signature_uncompressed", "compress_signature", "compress_pubkey", "aggregate_public_keys",
                 "aggregate_public_keys_subtract", "aggregate_signatures", "private_key_to_public_key", "generate_private_key",
                 "destroy_public_key", "destroy_signature", "destroy_private_key", "free_vec", "celo_amd_verify_hash"):
        getattr(lib, name).restype = C.c_bool
    return lib


def _take(lib, ptr, n):
    data = bytes(C.cast(ptr, C.POINTER(C.c_ubyte * n.value)).contents)
    assert lib.free_vec(ptr, n)
    return data


def _ser(lib, fn, handle):
    out, n = C.c_void_p(), C.c_int()
    assert getattr(lib, fn)(handle, C.byref(out), C.byref(n))
    return _take(lib, out, n)


def _deser(lib, fn, data):
    h = C.c_void_p()
    ok = getattr(lib, fn)(data, C.c_int(len(data)), C.byref(h))
    return h if ok else None


def test_header_symbols_exported(sys_lib):
    txt = open(os.path.join(ROOT, "include", "celo_bls_snark_sys.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name in sorted(set(re.findall(r"\bbool\s+(\w+)\s*\(", txt))):
        assert hasattr(sys_lib, name), name


@pytest.mark.parametrize("key", ["g1_compat", "g1_compat_cip22", "g1_noncompat"])
def test_signature_roundtrip_reference_points(sys_lib, golden, key):
    for hx in golden["hash_to_curve"][key]["points"]:
        b = bytes.fromhex(hx)
        h = _deser(sys_lib, "deserialize_signature", b)
        assert h is not None
        assert _ser(sys_lib, "serialize_signature", h) == b
        unc = _ser(sys_lib, "serialize_signature_uncompressed", h)
        P = ecc.deser_point(ecc.E1_377, b)
        assert unc == ecc.ser_point(ecc.E1_377, P, compressed=False)
        out, n = C.c_void_p(), C.c_int()
        assert sys_lib.compress_signature(unc, C.c_int(len(unc)), C.byref(out), C.byref(n))
        assert _take(sys_lib, out, n) == b
        assert sys_lib.destroy_signature(h)


def test_public_key_roundtrip_reference_points(sys_lib, golden):
    for hx in golden["hash_to_curve"]["g2_noncompat"]["points"]:
        b = bytes.fromhex(hx)
        h = _deser(sys_lib, "deserialize_public_key", b)
        assert h is not None
        assert _ser(sys_lib, "serialize_public_key", h) == b
        unc = _ser(sys_lib, "serialize_public_key_uncompressed", h)
        assert unc == ecc.ser_point(ecc.E2_377, ecc.deser_point(ecc.E2_377, b), compressed=False)
        out, n = C.c_void_p(), C.c_int()
        assert sys_lib.compress_pubkey(unc, C.c_int(len(unc)), C.byref(out), C.byref(n))
        assert _take(sys_lib, out, n) == b
        assert sys_lib.destroy_public_key(h)


def test_rejects_bad_encodings(sys_lib, golden):
    good = bytes.fromhex(golden["hash_to_curve"]["g1_compat"]["points"][0])
    # x >= q (non-canonical field element)
    bad = bytearray(ecc.Q377.to_bytes(48, "little"))
    assert _deser(sys_lib, "deserialize_signature", bytes(bad)) is None
    # x with no y on the curve
    x = 5
    while ecc.sqrt_fp((x ** 3 + 1) % ecc.Q377, ecc.Q377) is not None:
        x += 1
    assert _deser(sys_lib, "deserialize_signature", x.to_bytes(48, "little")) is None
    # on the curve but outside the prime-order subgroup (cofactor not cleared)
    x = 7
    while True:
        y = ecc.sqrt_fp((x ** 3 + 1) % ecc.Q377, ecc.Q377)
        if y is not None and not ecc.E1_377.in_subgroup((x, y)):
            break
        x += 1
    assert _deser(sys_lib, "deserialize_signature", ecc.ser_point(ecc.E1_377, (x, y))) is None
    # truncated input
    assert _deser(sys_lib, "deserialize_signature", good[:40]) is None
    # infinity round-trips
    inf = ecc.ser_point(ecc.E1_377, None)
    h = _deser(sys_lib, "deserialize_signature", inf)
    assert h is not None and _ser(sys_lib, "serialize_signature", h) == inf
    # sign AND infinity flags set: not an encoding (ark-serialize SWFlags::from_u8 returns None: the identity has one encoding)
    both = bytearray(inf)
    both[-1] |= 0x80
    assert _deser(sys_lib, "deserialize_signature", bytes(both)) is None
    both2 = bytearray(ecc.ser_point(ecc.E2_377, None))
    both2[-1] |= 0x80
    assert _deser(sys_lib, "deserialize_public_key", bytes(both2)) is None
    import numpy as np
    from oracle import cpu_oracle as co
    assert co.decompress("g1", bytes(both))[1].tolist() == [2] and co.decompress("g2", bytes(both2))[1].tolist() == [2]


def test_keys_and_aggregation(sys_lib):
    rng = ecc.SplitMix64(1234)
    sks = [ecc.random_scalar(rng, ecc.R377) for _ in range(4)]
    pk_handles, pk_points = [], []
    for sk in sks:
        skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
        assert skh is not None
        assert _ser(sys_lib, "serialize_private_key", skh) == sk.to_bytes(32, "little")
        pkh = C.c_void_p()
        assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
        exp = ecc.E2_377.mul(ecc.G2_377, sk)
        assert _ser(sys_lib, "serialize_public_key", pkh) == ecc.ser_point(ecc.E2_377, exp)
        pk_handles.append(pkh)
        pk_points.append(exp)
        assert sys_lib.destroy_private_key(skh)
    assert _deser(sys_lib, "deserialize_private_key", ecc.R377.to_bytes(32, "little")) is None   # sk must be < r
    arr = (C.c_void_p * 4)(*[h.value for h in pk_handles])
    agg = C.c_void_p()
    assert sys_lib.aggregate_public_keys(arr, C.c_int(4), C.byref(agg))
    total = None
    for P in pk_points:
        total = ecc.E2_377.add(total, P)
    assert _ser(sys_lib, "serialize_public_key", agg) == ecc.ser_point(ecc.E2_377, total)
    sub = C.c_void_p()
    arr2 = (C.c_void_p * 2)(pk_handles[1].value, pk_handles[3].value)
    assert sys_lib.aggregate_public_keys_subtract(agg, arr2, C.c_int(2), C.byref(sub))
    assert _ser(sys_lib, "serialize_public_key", sub) == ecc.ser_point(ecc.E2_377, ecc.E2_377.add(pk_points[0], pk_points[2]))
    # signatures: sum of 3 points, and the empty aggregate is the identity
    sig_pts = [ecc.E1_377.mul(ecc.G1_377, rng.next() | 1) for _ in range(3)]
    sh = [_deser(sys_lib, "deserialize_signature", ecc.ser_point(ecc.E1_377, P)) for P in sig_pts]
    arr3 = (C.c_void_p * 3)(*[h.value for h in sh])
    asig = C.c_void_p()
    assert sys_lib.aggregate_signatures(arr3, C.c_int(3), C.byref(asig))
    tot = None
    for P in sig_pts:
        tot = ecc.E1_377.add(tot, P)
    assert _ser(sys_lib, "serialize_signature", asig) == ecc.ser_point(ecc.E1_377, tot)
    empty = C.c_void_p()
    assert sys_lib.aggregate_signatures(None, C.c_int(0), C.byref(empty))
    assert _ser(sys_lib, "serialize_signature", empty) == ecc.ser_point(ecc.E1_377, None)
    gen = C.c_void_p()
    assert sys_lib.generate_private_key(C.byref(gen))
    assert int.from_bytes(_ser(sys_lib, "serialize_private_key", gen), "little") < ecc.R377


def test_aggregate_public_keys_has_the_cache_set_semantics(sys_lib):
    """aggregate_public_keys / aggregate_public_keys_subtract go through PublicKeyCache::aggregate in the reference
    (crates/bls-snark-sys/src/signatures.rs:428-474 -> crates/bls-crypto/src/bls/cache.rs:65-87): the keys are collected into a
    HashSet with byte equality of the Jacobian limbs (cache.rs:95-104), so a key listed twice counts ONCE and the order of the
    list does not matter; aggregate_signatures is a plain sum (signatures.rs:485-505), where a repeat counts twice."""
    rng = ecc.SplitMix64(4321)
    pts = [ecc.E2_377.mul(ecc.G2_377, ecc.random_scalar(rng, ecc.R377)) for _ in range(4)]
    hs = [_deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, P)) for P in pts]
    assert all(h is not None for h in hs)

    def agg(idx):
        arr = (C.c_void_p * len(idx))(*[hs[i].value for i in idx])
        out = C.c_void_p()
        assert sys_lib.aggregate_public_keys(arr, C.c_int(len(idx)), C.byref(out))
        return out

    def total(idx):
        t = None
        for i in idx:
            t = ecc.E2_377.add(t, pts[i])
        return ecc.ser_point(ecc.E2_377, t)

    assert _ser(sys_lib, "serialize_public_key", agg([0, 1, 2, 3])) == total([0, 1, 2, 3])
    assert _ser(sys_lib, "serialize_public_key", agg([0, 1, 1, 2, 0, 0])) == total([0, 1, 2])          # repeats count once
    assert _ser(sys_lib, "serialize_public_key", agg([2, 0, 1])) == total([0, 1, 2])                      # the same set in another order
    assert _ser(sys_lib, "serialize_public_key", agg([3, 3])) == total([3])
    # the same point behind two handles holding the same limbs (both came from the wire: Z = 1) is still one set element
    twin = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, pts[1]))
    arr = (C.c_void_p * 3)(hs[0].value, hs[1].value, twin.value)
    out = C.c_void_p()
    assert sys_lib.aggregate_public_keys(arr, C.c_int(3), C.byref(out))
    assert _ser(sys_lib, "serialize_public_key", out) == total([0, 1])
    # subtract: the list is aggregated as a set first, then subtracted once
    full = agg([0, 1, 2, 3])
    arr = (C.c_void_p * 4)(hs[1].value, hs[3].value, hs[1].value, hs[3].value)
    sub = C.c_void_p()
    assert sys_lib.aggregate_public_keys_subtract(full, arr, C.c_int(4), C.byref(sub))
    assert _ser(sys_lib, "serialize_public_key", sub) == total([0, 2])
    arr = (C.c_void_p * 2)(hs[3].value, hs[1].value)
    sub2 = C.c_void_p()
    assert sys_lib.aggregate_public_keys_subtract(full, arr, C.c_int(2), C.byref(sub2))
    assert _ser(sys_lib, "serialize_public_key", sub2) == total([0, 2])
    # signatures: a plain sum, a repeat counts twice
    S = ecc.E1_377.mul(ecc.G1_377, 12345)
    sh = _deser(sys_lib, "deserialize_signature", ecc.ser_point(ecc.E1_377, S))
    arr = (C.c_void_p * 2)(sh.value, sh.value)
    asig = C.c_void_p()
    assert sys_lib.aggregate_signatures(arr, C.c_int(2), C.byref(asig))
    assert _ser(sys_lib, "serialize_signature", asig) == ecc.ser_point(ecc.E1_377, ecc.E1_377.add(S, S))


@pytest.mark.gpu
def test_verify_hash_core_on_gpu(sys_lib, gpu):
    """verify_signature after hashing (crates/bls-snark-sys/src/signatures.rs:244 -> public.rs:94-120) on the GPU."""
    from oracle import cpu_oracle as co
    sk = 0x1F2E3D4C5B6A79887766554433221100
    Hm = ecc.E1_377.mul(ecc.G1_377, 0xABCDEF)
    sig = ecc.E1_377.mul(Hm, sk)
    pk = ecc.E2_377.mul(ecc.G2_377, sk)
    pkh = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, pk))
    sgh = _deser(sys_lib, "deserialize_signature", ecc.ser_point(ecc.E1_377, sig))
    hxy, _ = co.pack_g1_377([Hm])
    ok = C.c_bool(False)
    assert sys_lib.celo_amd_verify_hash(pkh, hxy.ctypes.data_as(C.c_void_p), sgh, C.byref(ok)) and ok.value
    bad = _deser(sys_lib, "deserialize_signature", ecc.ser_point(ecc.E1_377, ecc.E1_377.mul(Hm, sk + 1)))
    assert sys_lib.celo_amd_verify_hash(pkh, hxy.ctypes.data_as(C.c_void_p), bad, C.byref(ok)) and not ok.value


# ---------------------------------------------------------------- hashing / signing / verification with the DIRECT hasher
def _hash_direct(lib, msg, use_pop=False):
    out, n = C.c_void_p(), C.c_int()
    lib.hash_direct.restype = C.c_bool
    assert lib.hash_direct(msg, C.c_int(len(msg)), C.byref(out), C.byref(n), C.c_bool(use_pop))
    b = _take(lib, out, n)
    assert len(b) == 97 and b[96] == 0
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little"))


def test_hash_direct_matches_oracle_restatement(sys_lib):
    """hash_direct / hash_direct_with_attempt / hash_direct_first_step (crates/bls-snark-sys/src/signatures.rs:93-215) vs the
    oracle's Python restatement of DirectHasher + try-and-increment (itself pinned on the reference's Blake2 vectors)."""
    from oracle.py import hashing as hs
    msgs = [b"", b"hello", bytes(range(256)) * 3, b"\x00" * 64, b"celo epoch 1234"]
    for i, m in enumerate(msgs):
        for pop in (False, True):
            exp, c = hs.hash_to_g1_direct(b"ULforpop" if pop else b"ULforxof", m, b"")
            assert _hash_direct(sys_lib, m, pop) == exp
            out, n, att = C.c_void_p(), C.c_int(), C.c_int(-1)
            sys_lib.hash_direct_with_attempt.restype = C.c_bool
            assert sys_lib.hash_direct_with_attempt(m, C.c_int(len(m)), C.byref(out), C.byref(n), C.byref(att), C.c_bool(pop))
            _take(sys_lib, out, n)
            assert att.value == c
        out, n = C.c_void_p(), C.c_int()
        sys_lib.hash_direct_first_step.restype = C.c_bool
        assert sys_lib.hash_direct_first_step(m, C.c_int(len(m)), C.c_int(33 + 31 * i), C.byref(out), C.byref(n))
        assert _take(sys_lib, out, n) == hs.direct_hash(b"ULforxof", m, 33 + 31 * i)


def test_sign_message_is_sk_times_hash(sys_lib):
    from oracle.py import hashing as hs
    sk = 0x0123456789ABCDEF00112233445566778899AABBCCDDEEFF0011223344 % ecc.R377
    skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
    sys_lib.sign_message.restype = C.c_bool
    sys_lib.sign_pop.restype = C.c_bool
    sig = C.c_void_p()
    msg, extra = b"message", b"extra-data"
    assert sys_lib.sign_message(skh, msg, C.c_int(len(msg)), extra, C.c_int(len(extra)), C.c_bool(False), C.c_bool(False), C.byref(sig))
    H, _ = hs.hash_to_g1_direct(b"ULforxof", msg, extra)
    assert _ser(sys_lib, "serialize_signature", sig) == ecc.ser_point(ecc.E1_377, ecc.E1_377.mul(H, sk))
    pop = C.c_void_p()
    assert sys_lib.sign_pop(skh, msg, C.c_int(len(msg)), C.byref(pop))
    Hp, _ = hs.hash_to_g1_direct(b"ULforpop", msg, b"")
    assert _ser(sys_lib, "serialize_signature", pop) == ecc.ser_point(ecc.E1_377, ecc.E1_377.mul(Hp, sk))
    # composite hasher (plain and cip22): sigma = sk * H_composite(message)
    for cip22 in (False, True):
        csig = C.c_void_p()
        assert sys_lib.sign_message(skh, msg, C.c_int(len(msg)), extra, C.c_int(len(extra)), C.c_bool(True), C.c_bool(cip22), C.byref(csig))
        Hc, _ = hs.hash_to_g1(b"ULforxof", msg, extra, composite=True, cip22=cip22)
        assert _ser(sys_lib, "serialize_signature", csig) == ecc.ser_point(ecc.E1_377, ecc.E1_377.mul(Hc, sk))
    # (composite=false, cip22=true) is an error in the reference as well (signatures.rs:61)
    assert not sys_lib.sign_message(skh, msg, C.c_int(len(msg)), extra, C.c_int(len(extra)), C.c_bool(False), C.c_bool(True), C.byref(sig))


class _Buffer(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_size_t)]


class _MessageFFI(C.Structure):
    _fields_ = [("data", _Buffer), ("extra", _Buffer), ("public_key", C.c_void_p), ("sig", C.c_void_p)]


class _BatchMessageFFI(C.Structure):
    _fields_ = [("data", _Buffer), ("extra", _Buffer), ("public_keys", C.POINTER(C.c_void_p)), ("public_keys_len", C.c_size_t),
                ("signatures", C.POINTER(C.c_void_p)), ("signatures_len", C.c_size_t)]


@pytest.mark.gpu
@pytest.mark.parametrize("composite,cip22", [(False, False), (True, False), (True, True)])
def test_sign_verify_flow_on_gpu(sys_lib, gpu, composite, cip22):
    """The reference's randomised self-consistency tests (crates/bls-crypto/src/bls/signature.rs:180-426) through the C ABI:
    sign -> verify OK; wrong message / key -> not verified; PoP; batch_verify over epochs; strict batches accept then reject —
    with the direct hasher and with the composite hasher before / after CIP22 (signatures.rs:45-400 flag combinations)."""
    CF, C22 = C.c_bool(composite), C.c_bool(cip22)
    for f in ("sign_message", "sign_pop", "verify_signature", "verify_pop", "batch_verify_signature", "batch_verify_strict"):
        getattr(sys_lib, f).restype = C.c_bool
    rng = ecc.SplitMix64(2024)

    def keypair():
        sk = ecc.random_scalar(rng, ecc.R377)
        skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
        pkh = C.c_void_p()
        assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
        return skh, pkh

    def sign(skh, msg, extra=b""):
        s = C.c_void_p()
        assert sys_lib.sign_message(skh, msg, C.c_int(len(msg)), extra, C.c_int(len(extra)), CF, C22, C.byref(s))
        return s

    sk1, pk1 = keypair()
    sk2, pk2 = keypair()
    msg, extra = b"hello", b"extra"
    s1 = sign(sk1, msg, extra)
    ok = C.c_bool(False)
    assert sys_lib.verify_signature(pk1, msg, C.c_int(5), extra, C.c_int(5), s1, CF, C22, C.byref(ok)) and ok.value
    assert sys_lib.verify_signature(pk1, b"hellp", C.c_int(5), extra, C.c_int(5), s1, CF, C22, C.byref(ok)) and not ok.value
    assert sys_lib.verify_signature(pk2, msg, C.c_int(5), extra, C.c_int(5), s1, CF, C22, C.byref(ok)) and not ok.value
    assert not sys_lib.verify_signature(pk1, msg, C.c_int(5), extra, C.c_int(5), s1, C.c_bool(False), C.c_bool(True), C.byref(ok))
    pop = C.c_void_p()
    pkb = _ser(sys_lib, "serialize_public_key", pk1)
    assert sys_lib.sign_pop(sk1, pkb, C.c_int(len(pkb)), C.byref(pop))
    assert sys_lib.verify_pop(pk1, pkb, C.c_int(len(pkb)), pop, C.byref(ok)) and ok.value
    assert sys_lib.verify_pop(pk2, pkb, C.c_int(len(pkb)), pop, C.byref(ok)) and not ok.value

    # batch_verify_signature: 4 epochs, each with an aggregate key of 3 validators
    msgs = []
    keep = []
    for e in range(4):
        m = b"epoch-%d" % e
        ks = [keypair() for _ in range(3)]
        sigs = [sign(sk, m, b"x") for sk, _ in ks]
        pk_arr = (C.c_void_p * 3)(*[pk.value for _, pk in ks])
        apk = C.c_void_p()
        assert sys_lib.aggregate_public_keys(pk_arr, C.c_int(3), C.byref(apk))
        sg_arr = (C.c_void_p * 3)(*[s.value for s in sigs])
        asig = C.c_void_p()
        assert sys_lib.aggregate_signatures(sg_arr, C.c_int(3), C.byref(asig))
        keep.append((m, apk, asig, ks, sigs))
        msgs.append(_MessageFFI(_Buffer(m, len(m)), _Buffer(b"x", 1), apk.value, asig.value))
    arr = (_MessageFFI * 4)(*msgs)
    assert sys_lib.batch_verify_signature(arr, C.c_size_t(4), CF, C22, C.byref(ok)) and ok.value
    msgs[2] = _MessageFFI(_Buffer(b"epoch-9", 7), _Buffer(b"x", 1), keep[2][1].value, keep[2][2].value)
    arr = (_MessageFFI * 4)(*msgs)
    assert sys_lib.batch_verify_signature(arr, C.c_size_t(4), CF, C22, C.byref(ok)) and not ok.value

    # batch_verify_strict: 3 batches; the second contains one signature on the wrong message
    batches, holders = [], []
    for b in range(3):
        m = b"strict-%d" % b
        ks = [keypair() for _ in range(4)]
        sigs = [sign(sk, m) for sk, _ in ks]
        if b == 1:
            sigs[2] = sign(ks[2][0], b"other")
        pks = (C.c_void_p * 4)(*[pk.value for _, pk in ks])
        sgs = (C.c_void_p * 4)(*[s.value for s in sigs])
        holders.append((pks, sgs, m))
        batches.append(_BatchMessageFFI(_Buffer(m, len(m)), _Buffer(b"", 0), pks, 4, sgs, 4))
    barr = (_BatchMessageFFI * 3)(*batches)
    res = (C.c_bool * 3)()
    assert not sys_lib.batch_verify_strict(barr, C.c_size_t(3), CF, C22, res)
    assert list(res) == [True, False, True]
    good = (_BatchMessageFFI * 2)(batches[0], batches[2])
    res2 = (C.c_bool * 2)()
    assert sys_lib.batch_verify_strict(good, C.c_size_t(2), CF, C22, res2) and list(res2) == [True, True]
    # Batch::verify zips keys with signatures (crates/bls-crypto/src/bls/batch.rs:60-64): a batch that lists 4 keys and only their
    # first 3 signatures is checked on those 3 pairs; the other batch of the call is unaffected
    pks0, sgs0, m0 = holders[0]
    short = _BatchMessageFFI(_Buffer(m0, len(m0)), _Buffer(b"", 0), pks0, 4, sgs0, 3)
    mixed = (_BatchMessageFFI * 2)(short, batches[2])
    res3 = (C.c_bool * 2)(False, False)
    assert sys_lib.batch_verify_strict(mixed, C.c_size_t(2), CF, C22, res3) and list(res3) == [True, True]
    # out_results is written on every path: a call that fails as a whole leaves "not verified" everywhere
    res4 = (C.c_bool * 2)(True, True)
    assert not sys_lib.batch_verify_strict(mixed, C.c_size_t(2), C.c_bool(False), C.c_bool(True), res4) and list(res4) == [False, False]
    assert not sys_lib.batch_verify_strict(good, C.c_size_t(2), C.c_bool(False), C.c_bool(True), res2) and list(res2) == [False, False]


@pytest.mark.gpu
def test_batch_verify_signature_hashes_many_messages_on_the_gpu(sys_lib, gpu):
    """From 256 messages up, batch_verify_signature / batch_verify_strict with the direct hasher hash all messages in one GPU
    call (hash_direct.h) instead of on the host cores: 300 epochs of varying message length accept; one changed message byte,
    one changed extra-data byte reject - the same verdicts as the per-message host path gives on the first 8."""
    for f in ("sign_message", "batch_verify_signature", "batch_verify_strict"):
        getattr(sys_lib, f).restype = C.c_bool
    rng = ecc.SplitMix64(77)
    CF, C22 = C.c_bool(False), C.c_bool(False)
    sk = ecc.random_scalar(rng, ecc.R377)
    skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
    pkh = C.c_void_p()
    assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
    n = 300
    msgs = [bytes([(7 * i + j) & 0xFF for j in range(1 + (i * 5) % 150)]) for i in range(n)]
    extras = [bytes([i & 0xFF]) * (i % 3) for i in range(n)]
    sigs = []
    for m, e in zip(msgs, extras):
        s = C.c_void_p()
        assert sys_lib.sign_message(skh, m, C.c_int(len(m)), e, C.c_int(len(e)), CF, C22, C.byref(s))
        sigs.append(s)

    def run(ms, es, count):
        arr = (_MessageFFI * count)(*[_MessageFFI(_Buffer(ms[i], len(ms[i])), _Buffer(es[i], len(es[i])), pkh.value, sigs[i].value) for i in range(count)])
        ok = C.c_bool(False)
        assert sys_lib.batch_verify_signature(arr, C.c_size_t(count), CF, C22, C.byref(ok))
        return ok.value

    assert run(msgs, extras, n) and run(msgs, extras, 8)
    bad = list(msgs)
    bad[257] = bad[257][:-1] + bytes([bad[257][-1] ^ 1])
    assert not run(bad, extras, n)
    bade = list(extras)
    bade[5] = b"\x99"
    assert not run(msgs, bade, n) and not run(msgs, bade, 8)
    # strict batches: 256 one-signer batches, batch 100 signed over a different message
    keys = (C.c_void_p * 1)(pkh.value)
    holders, batches = [], []
    for i in range(256):
        sg = (C.c_void_p * 1)(sigs[i if i != 100 else 101].value)
        holders.append(sg)
        batches.append(_BatchMessageFFI(_Buffer(msgs[i], len(msgs[i])), _Buffer(extras[i], len(extras[i])), keys, 1, sg, 1))
    res = (C.c_bool * 256)()
    assert not sys_lib.batch_verify_strict((_BatchMessageFFI * 256)(*batches), C.c_size_t(256), CF, C22, res)
    assert [i for i in range(256) if not res[i]] == [100]


@pytest.mark.gpu
def test_cip22_tail_on_gpu_reproduces_reference_points(sys_lib, gpu, golden):
    """hash_to_g1_cip22_tail_bls12_377 (the CIP22 try-and-increment loop on the GPU, inner CRHs from hash_crh) on the inputs of
    the reference's compat CIP22 vectors (crates/bls-crypto/src/hash_to_curve/mod.rs:438-449): the ten reference points come
    out byte for byte, with the attempt counters of the host path."""
    from oracle import cpu_oracle as co
    gen, _ = _composite_inputs()
    lib = sys_lib
    lib.hash_crh.restype = C.c_bool
    lib.celo_amd_hash_to_g1.restype = C.c_bool
    pts = golden["hash_to_curve"]["g1_compat_cip22"]["points"]
    doms, inners, extras, want_att = [], [], [], []
    for dom, msg, extra in gen(len(pts)):
        out, n = C.c_void_p(), C.c_int()
        assert lib.hash_crh(msg, len(msg), 96, C.byref(out), C.byref(n))
        inners.append(_take(lib, out, n))
        doms.append(dom); extras.append(extra)
        o48, att = (C.c_ubyte * 48)(), C.c_int(-1)
        assert lib.celo_amd_hash_to_g1(True, True, dom, msg, len(msg), extra, len(extra), o48, C.byref(att))
        want_att.append(att.value)
    for i, hx in enumerate(pts):                   # the reference draws a fresh 8-byte domain per vector: one call each
        xy, att = gpu.hash_to_g1_direct(doms[i], [inners[i]] * 3, [extras[i]] * 3, cip22_tail=True)
        assert att.tolist() == [want_att[i]] * 3
        vals = co.from_mont(xy.reshape(-1, 6), ecc.Q377)
        for k in range(3):
            assert ecc.ser_point(ecc.E1_377, (vals[2 * k], vals[2 * k + 1])).hex() == hx


@pytest.mark.gpu
@pytest.mark.parametrize("key,cip22", [("g1_compat", False), ("g1_compat_cip22", True)])
def test_composite_hash_to_g1_on_gpu_reproduces_reference_points(gpu, golden, key, cip22):
    """hash_to_g1_composite_bls12_377 - Pedersen CRH, XOF, try-and-increment and cofactor all on the GPU - on the inputs of the
    reference's compat vectors (crates/bls-crypto/src/hash_to_curve/mod.rs:412-449): the reference's twenty points byte for
    byte, before and after CIP22."""
    from oracle import cpu_oracle as co
    gen, _ = _composite_inputs()
    pts = golden["hash_to_curve"][key]["points"]
    for (dom, msg, extra), hx in zip(gen(len(pts)), pts):
        xy, att = gpu.hash_to_g1_composite(dom, [msg, msg], [extra, extra], cip22=cip22)
        assert att[0] == att[1] < 255
        vals = co.from_mont(xy.reshape(-1, 6), ecc.Q377)
        assert ecc.ser_point(ecc.E1_377, (vals[0], vals[1])).hex() == hx and vals[2:] == vals[:2]


@pytest.mark.gpu
def test_bulk_pedersen_crh_on_gpu(sys_lib, gpu, golden):
    """composite_crh_bls12_377 (pedersen.h, one message per lane) against the reference's CRH vector
    (crates/bls-crypto/src/hashers/composite.rs:105-190, golden "composite_hasher" entries without an XOF length) and against
    hash_crh, the one-message host path of the same source, for lengths 0 ... 400 (chunk boundaries fall on every bit offset;
    the last chunk is padded with zero bits)."""
    _, xs = _composite_inputs()
    lib = sys_lib
    lib.hash_crh.restype = C.c_bool
    ref = [(xs(v["seed0"], v["msg_len"]) if v["seed0"] is not None else b"", v["expected"])
           for v in golden["composite_hasher"].values() if v["out_bytes"] is None]
    assert ref
    got = gpu.composite_crh([m for m, _ in ref])
    assert [g.hex() for g in got] == [e for _, e in ref]
    msgs = [bytes([(11 * i + 3 * j) & 0xFF for j in range(l)]) for i, l in enumerate(list(range(0, 70)) + [95, 96, 97, 200, 333, 400])]
    want = []
    for m in msgs:
        out, n = C.c_void_p(), C.c_int()
        assert lib.hash_crh(m, len(m), 96, C.byref(out), C.byref(n))
        want.append(_take(lib, out, n))
    assert gpu.composite_crh(msgs) == want
    assert gpu.composite_crh([]) == []
    with pytest.raises(RuntimeError):
        gpu.composite_crh([bytes(93 * 560 * 3 // 8 + 1)])     # longer than the generator table: the reference panics


@pytest.mark.gpu
@pytest.mark.parametrize("cip22", [False, True])
def test_batch_verify_signature_composite_many_messages(sys_lib, gpu, cip22):
    """300 messages with the composite hasher, before and after CIP22: all hashing (Pedersen CRHs, try-and-increment rounds) on
    the GPU; accept, and reject after one changed message byte."""
    for f in ("sign_message", "batch_verify_signature"):
        getattr(sys_lib, f).restype = C.c_bool
    rng = ecc.SplitMix64(78)
    CF, C22 = C.c_bool(True), C.c_bool(cip22)
    sk = ecc.random_scalar(rng, ecc.R377)
    skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
    pkh = C.c_void_p()
    assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
    n = 300
    msgs = [bytes([(3 * i + j) & 0xFF for j in range(1 + (i * 7) % 90)]) for i in range(n)]
    extras = [bytes([i & 0xFF]) * (i % 4) for i in range(n)]
    sigs = []
    for m, e in zip(msgs, extras):
        s = C.c_void_p()
        assert sys_lib.sign_message(skh, m, C.c_int(len(m)), e, C.c_int(len(e)), CF, C22, C.byref(s))
        sigs.append(s)

    def run(ms):
        arr = (_MessageFFI * n)(*[_MessageFFI(_Buffer(ms[i], len(ms[i])), _Buffer(extras[i], len(extras[i])), pkh.value, sigs[i].value) for i in range(n)])
        ok = C.c_bool(False)
        assert sys_lib.batch_verify_signature(arr, C.c_size_t(n), CF, C22, C.byref(ok))
        return ok.value

    assert run(msgs)
    bad = list(msgs)
    bad[299] = bytes([bad[299][0] ^ 0x80]) + bad[299][1:]
    assert not run(bad)


# ---------------------------------------------------------------- snark half of the FFI
class _EpochBlockFFI(C.Structure):
    _fields_ = [("index", C.c_uint16), ("round", C.c_uint8), ("epoch_entropy", C.c_char_p), ("parent_entropy", C.c_char_p),
                ("pubkeys", C.c_char_p), ("pubkeys_num", C.c_size_t), ("maximum_non_signers", C.c_uint32),
                ("maximum_validators", C.c_size_t)]


def test_encode_epoch_block_symbols(sys_lib, golden):
    """encode_epoch_block_to_bytes reproduces the reference's pre-Donut golden encoding
    (crates/epoch-snark/src/epoch_block.rs:246,287-299) byte for byte; the CIP22 inner encoding matches the oracle restatement."""
    from oracle.py import epoch as ep
    gen = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, ecc.G2_377))
    arr = (C.c_void_p * 10)(*[gen.value] * 10)
    out, n = C.c_void_p(), C.c_int()
    sys_lib.encode_epoch_block_to_bytes.restype = C.c_bool
    assert sys_lib.encode_epoch_block_to_bytes(C.c_ushort(120), C.c_uint(3), arr, C.c_int(10), C.byref(out), C.byref(n))
    assert _take(sys_lib, out, n).hex() == golden["epoch_encoding"]["EXPECTED_ENCODING_BEFORE_DONUT"]
    sys_lib.encode_epoch_block_to_bytes_cip22.restype = C.c_bool
    o1, n1, o2, n2 = C.c_void_p(), C.c_int(), C.c_void_p(), C.c_int()
    e_ent, p_ent = bytes([7] * 16), bytes([9] * 16)
    assert sys_lib.encode_epoch_block_to_bytes_cip22(C.c_ushort(120), C.c_ubyte(5), e_ent, p_ent, C.c_uint(3), C.c_uint(12), arr, C.c_int(10),
                                                     C.byref(o1), C.byref(n1), C.byref(o2), C.byref(n2))
    inner, extra = ep.encode_inner_to_bytes_cip22(ep.EpochBlock(120, 5, e_ent, p_ent, 3, 12, [ecc.G2_377] * 10))
    assert _take(sys_lib, o1, n1) == inner and _take(sys_lib, o2, n2) == extra


def test_encode_epoch_block_with_an_identity_validator_key(sys_lib):
    """the reference's `read_pubkeys` (crates/bls-snark-sys/src/snark/epoch_block.rs:187-196) and `encode_public_key`
    (crates/epoch-snark/src/encoding.rs:23-47) take the identity as a validator key - arkworks' GroupAffine::zero() = (0, 1, infinity): 754 zero
    bits and a clear sign bit - where rounds 2-5 of this library refused it.  Both encoders now emit what the restatement of those lines emits."""
    from oracle.py import epoch as ep
    gen = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, ecc.G2_377))
    two = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, ecc.E2_377.mul(ecc.G2_377, 2)))
    ident = _deser(sys_lib, "deserialize_public_key", ecc.ser_point(ecc.E2_377, None))
    handles = [gen.value, ident.value, two.value, ident.value]
    keys = [ecc.G2_377, None, ecc.E2_377.mul(ecc.G2_377, 2), None]
    arr = (C.c_void_p * 4)(*handles)
    out, n = C.c_void_p(), C.c_int()
    sys_lib.encode_epoch_block_to_bytes.restype = C.c_bool
    assert sys_lib.encode_epoch_block_to_bytes(C.c_ushort(7), C.c_uint(1), arr, C.c_int(4), C.byref(out), C.byref(n))
    got = _take(sys_lib, out, n)
    assert got == ep.EpochBlock(7, 0, None, None, 1, 4, keys).encode_to_bytes()
    bits = ep.EpochBlock(7, 0, None, None, 1, 4, keys).encode_to_bits()
    assert bits[48 + 755:48 + 2 * 755] == [0] * 755                       # the identity's slot: x = 0, sign bit clear
    sys_lib.encode_epoch_block_to_bytes_cip22.restype = C.c_bool
    o1, n1, o2, n2 = C.c_void_p(), C.c_int(), C.c_void_p(), C.c_int()
    e_ent, p_ent = bytes([3] * 16), bytes([4] * 16)
    assert sys_lib.encode_epoch_block_to_bytes_cip22(C.c_ushort(7), C.c_ubyte(2), e_ent, p_ent, C.c_uint(1), C.c_uint(5), arr, C.c_int(4),
                                                     C.byref(o1), C.byref(n1), C.byref(o2), C.byref(n2))
    inner, extra = ep.encode_inner_to_bytes_cip22(ep.EpochBlock(7, 2, e_ent, p_ent, 1, 5, keys))
    assert _take(sys_lib, o1, n1) == inner and _take(sys_lib, o2, n2) == extra


@pytest.mark.gpu
def test_groth16_verify_takes_an_identity_validator_key(sys_lib, gpu, golden):
    """... and `verify` decodes such a block and runs the check (the reference's vector with one validator key replaced by the identity: the
    public inputs change, so the proof is rejected - by the pairing check, not by the decoder), also when the whole set sums to the identity."""
    from oracle.py import epoch as ep
    g = golden["groth16_bw6_761"]
    vk, proof = bytes.fromhex(g["vk"]), bytes.fromhex(g["proof"])
    fp, lp = bytes.fromhex(g["first_pubkeys"]), bytes.fromhex(g["last_pubkeys"])
    fe, fpe = bytes.fromhex(g["first_epoch_entropy"]), bytes.fromhex(g["first_parent_entropy"])
    le, lpe = bytes.fromhex(g["last_epoch_entropy"]), bytes.fromhex(g["last_parent_entropy"])

    def blk(d, pk, ee, pe):
        return _EpochBlockFFI(d["index"], d["round"], ee, pe, pk, d["pubkeys_num"], d["maximum_non_signers"], d["maximum_validators"])

    sys_lib.verify.restype = C.c_bool
    sys_lib.verify.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, _EpochBlockFFI, _EpochBlockFFI]
    assert sys_lib.verify(vk, len(vk), proof, len(proof), blk(g["first"], fp, fe, fpe), blk(g["last"], lp, le, lpe)) is True
    ident = ecc.ser_point(ecc.E2_377, None)
    lp_id = ident + lp[96:]
    assert sys_lib.verify(vk, len(vk), proof, len(proof), blk(g["first"], fp, fe, fpe), blk(g["last"], lp_id, le, lpe)) is False
    all_id = ident * g["last"]["pubkeys_num"]
    assert sys_lib.verify(vk, len(vk), proof, len(proof), blk(g["first"], fp, fe, fpe), blk(g["last"], all_id, le, lpe)) is False
    # the restatement encodes the same blocks (what the rejected hashes were computed over)
    keys = [None] * g["last"]["pubkeys_num"]
    blk_py = ep.EpochBlock(g["last"]["index"], g["last"]["round"], le, lpe, g["last"]["maximum_non_signers"], g["last"]["maximum_validators"], keys)
    assert len(blk_py.encode_last_epoch_to_bytes_with_aggregated_pk_cip22()) > 0


@pytest.mark.gpu
def test_reference_groth16_ffi_test_passes_on_gpu(sys_lib, gpu, golden):
    """The reference's own FFI test `simple_verifier_groth16_with_entropy` (crates/bls-snark-sys/src/snark/mod.rs:52-119),
    replayed against this library's `verify` symbol: must return true; flipping a byte of the proof or of an entropy value
    must return false."""
    g = golden["groth16_bw6_761"]
    vk, proof = bytes.fromhex(g["vk"]), bytes.fromhex(g["proof"])
    fp, lp = bytes.fromhex(g["first_pubkeys"]), bytes.fromhex(g["last_pubkeys"])

    def blk(d, pk, ee, pe):
        return _EpochBlockFFI(d["index"], d["round"], ee, pe, pk, d["pubkeys_num"], d["maximum_non_signers"], d["maximum_validators"])

    fe, fpe = bytes.fromhex(g["first_epoch_entropy"]), bytes.fromhex(g["first_parent_entropy"])
    le, lpe = bytes.fromhex(g["last_epoch_entropy"]), bytes.fromhex(g["last_parent_entropy"])
    sys_lib.verify.restype = C.c_bool
    sys_lib.verify.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, _EpochBlockFFI, _EpochBlockFFI]
    first, last = blk(g["first"], fp, fe, fpe), blk(g["last"], lp, le, lpe)
    assert sys_lib.verify(vk, len(vk), proof, len(proof), first, last) is True
    bad_entropy = bytes([le[0] ^ 1]) + le[1:]
    assert sys_lib.verify(vk, len(vk), proof, len(proof), first, blk(g["last"], lp, bad_entropy, lpe)) is False
    assert sys_lib.verify(vk, len(vk), proof, len(proof), blk(g["first"], fp, fe, bytes([fpe[0] ^ 1]) + fpe[1:]), last) is False
    # a different (valid) G1 point as the proof's C: swap A and C
    swapped = proof[192:288] + proof[96:192] + proof[0:96]
    assert sys_lib.verify(vk, len(vk), swapped, len(swapped), first, last) is False


# ---------------------------------------------------------------- composite (Bowe-Hopwood) hasher through the C ABI
def _composite_inputs():
    from tests.test_oracle_golden import reference_hash_test_inputs, _xorshift_bytes
    return reference_hash_test_inputs, _xorshift_bytes


def test_composite_hasher_reference_vectors(sys_lib, golden):
    """crates/bls-crypto/src/hashers/composite.rs:105-190 through hash_crh (signatures.rs:169) and the XOF hook."""
    _, xs = _composite_inputs()
    lib = sys_lib
    lib.hash_crh.restype = C.c_bool
    lib.celo_amd_composite_hash.restype = C.c_bool
    for name, v in golden["composite_hasher"].items():
        msg = xs(v["seed0"], v["msg_len"]) if v["seed0"] is not None else b""
        if v["out_bytes"] is None:
            out, n = C.c_void_p(), C.c_int()
            assert lib.hash_crh(msg, len(msg), 96, C.byref(out), C.byref(n))
            assert n.value == 48
            got = _take(lib, out, n)
        else:
            buf = (C.c_ubyte * v["out_bytes"])()
            assert lib.celo_amd_composite_hash(b"ULforxof", msg, len(msg), v["out_bytes"], buf)
            got = bytes(buf)
        assert got.hex() == v["expected"], name
    # test_invalid_message (composite.rs:193): a 1 MB message exceeds 93 * 560 * 3 bits -> error, not a crash
    big = bytes(1_000_000)
    out, n = C.c_void_p(), C.c_int()
    assert not lib.hash_crh(big, len(big), 96, C.byref(out), C.byref(n))


@pytest.mark.parametrize("key,cip22", [("g1_compat", False), ("g1_compat_cip22", True)])
def test_composite_hash_to_g1_reference_vectors(sys_lib, golden, key, cip22):
    """hash_to_curve/mod.rs:412-455 through the product's try-and-increment (both variants, compat bit logic)."""
    gen, _ = _composite_inputs()
    lib = sys_lib
    lib.celo_amd_hash_to_g1.restype = C.c_bool
    pts = golden["hash_to_curve"][key]["points"]
    for (dom, msg, extra), hx in zip(gen(len(pts)), pts):
        out = (C.c_ubyte * 48)()
        att = C.c_int(-1)
        assert lib.celo_amd_hash_to_g1(True, cip22, dom, msg, len(msg), extra, len(extra), out, C.byref(att))
        assert bytes(out).hex() == hx
        assert 0 <= att.value < 255


def test_hash_composite_symbols(sys_lib):
    """hash_composite / hash_composite_cip22 (signatures.rs:143,215): 144-byte projective ToBytes of the SIG_DOMAIN hash; the point
    equals the oracle's hash and the attempt counter is reported."""
    from oracle.py import hashing as hs
    lib = sys_lib
    lib.hash_composite.restype = C.c_bool
    lib.hash_composite_cip22.restype = C.c_bool
    msg, extra = b"composite message", b"\x01\x02"
    for cip22 in (False, True):
        out, n = C.c_void_p(), C.c_int()
        att = C.c_ubyte(255)
        if cip22:
            assert lib.hash_composite_cip22(msg, len(msg), extra, len(extra), C.byref(out), C.byref(n), C.byref(att))
        else:
            assert lib.hash_composite(msg, len(msg), extra, len(extra), C.byref(out), C.byref(n))
        assert n.value == 144
        b = _take(lib, out, n)
        x, y, z = (int.from_bytes(b[i:i + 48], "little") for i in (0, 48, 96))
        zi = pow(z, -1, ecc.Q377)
        P, c, pre = hs.hash_to_g1(b"ULforxof", msg, extra, composite=True, cip22=cip22, want_pre=True)
        assert (x * zi * zi % ecc.Q377, y * zi * zi * zi % ecc.Q377) == tuple(P)
        # round 5: the bytes are those of arkworks' own Jacobian representative (scale_by_cofactor's schedule, oracle/py/hashing.py), z != 1
        assert (x, y, z) == hs.ark_scale_by_cofactor_jacobian(pre) and z != 1
        if cip22:
            assert att.value == c


def test_direct_hasher_random_vectors(sys_lib, golden):
    """crates/bls-crypto/src/hashers/direct.rs:99-147 through the product's DirectHasher (crh, xof, hash), explicit domains."""
    from tests.test_oracle_golden import _xorshift_bytes
    lib = sys_lib
    lib.celo_amd_direct_hasher.restype = C.c_bool

    def run(what, dom, msg, n, outlen):
        buf = (C.c_ubyte * outlen)()
        assert lib.celo_amd_direct_hasher(what, dom, len(dom), msg, len(msg), n, buf)
        return bytes(buf)

    for name, v in golden["direct_hasher_random"].items():
        msg = _xorshift_bytes(v["seed0"], v["msg_len"])
        if name == "test_crh_random":
            got = run(0, b"", msg, 96, 32)
        elif name == "test_xof_random_96":
            got = run(1, b"ULforxof", run(0, b"", msg, 96, 32), 96, 96)
        else:
            got = run(2, b"ULforxof", msg, 96, 96)
        assert got.hex() == v["expected"], name


def test_cached_public_key_deserialisation(sys_lib, golden):
    """crates/bls-crypto/src/bls/cache.rs:124-160 (`deserializer`, `caches_deserialized_pubkeys`): the cached entry point returns
    the same key as the plain one, hit or miss, far more keys than the 512-entry LRU holds, and bad encodings still fail."""
    import time
    lib = sys_lib
    lib.deserialize_public_key_cached.restype = C.c_bool
    pts = [bytes.fromhex(h) for h in golden["hash_to_curve"]["g2_noncompat"]["points"]]
    for rounds in range(2):                                  # second round = cache hits
        for b in pts:
            h1, h2 = C.c_void_p(), C.c_void_p()
            assert lib.deserialize_public_key(b, len(b), C.byref(h1)) and lib.deserialize_public_key_cached(b, len(b), C.byref(h2))
            assert _ser(lib, "serialize_public_key", h1) == _ser(lib, "serialize_public_key", h2) == b
            assert lib.destroy_public_key(h1) and lib.destroy_public_key(h2)
    t0 = time.perf_counter()
    for _ in range(200):
        h = C.c_void_p()
        assert lib.deserialize_public_key_cached(pts[0], 96, C.byref(h))
        lib.destroy_public_key(h)
    hit = (time.perf_counter() - t0) / 200
    assert hit < 2e-4                                        # a hit is a copy, not a square root + subgroup check (~1 ms)
    bad = bytes([pts[0][0] ^ 1]) + pts[0][1:]
    hb = C.c_void_p()
    ok = lib.deserialize_public_key_cached(bad, 96, C.byref(hb))
    assert ok == lib.deserialize_public_key(bad, 96, C.byref(C.c_void_p()))


@pytest.mark.gpu
def test_concurrent_verifications_are_combined(sys_lib, gpu):
    """bls-snark-sys is synchronous and re-entrant: 32 host threads verify at once; the library
    combines their single-product checks into shared launches.  Every verdict must be right (valid / wrong message mixed),
    and the wall time must show the combining (far below 32 x one call)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    lib = sys_lib
    for f in ("sign_message", "verify_signature"):
        getattr(lib, f).restype = C.c_bool
    sk = _deser(lib, "deserialize_private_key", (0xABCDEF123456789 % ecc.R377).to_bytes(32, "little"))
    pk = C.c_void_p()
    assert lib.private_key_to_public_key(sk, C.byref(pk))
    msgs = [b"message-%03d" % i for i in range(16)]
    sigs = []
    for m in msgs:
        s = C.c_void_p()
        assert lib.sign_message(sk, m, C.c_int(len(m)), b"", C.c_int(0), C.c_bool(False), C.c_bool(False), C.byref(s))
        sigs.append(s)

    def one(i):
        m = msgs[i % 16]
        claimed = m if i % 5 else b"message-xxx"          # every fifth call verifies the wrong message
        ok = C.c_bool(False)
        assert lib.verify_signature(pk, claimed, C.c_int(len(claimed)), b"", C.c_int(0), sigs[i % 16], C.c_bool(False), C.c_bool(False), C.byref(ok))
        return ok.value == bool(i % 5)

    assert one(1) and one(5)                               # warm-up, serial
    t0 = time.perf_counter()
    assert one(2)
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(32) as ex:
        res = list(ex.map(one, range(256)))
    t_all = time.perf_counter() - t0
    assert all(res)
    print("one call %.1f ms; 256 calls from 32 threads %.1f ms" % (t_one * 1e3, t_all * 1e3))
    assert t_all < 0.35 * 256 * t_one


@pytest.mark.gpu
def test_batch_verify_strict_oversized_and_empty_batches(sys_lib, gpu):
    """Batch::verify takes any number of signers and accepts an empty batch (crates/bls-crypto/src/bls/batch.rs:44-84).  A batch
    beyond the per-workgroup batched-MSM path (1024 signers) makes the whole call take the big MSM pipeline instance by instance;
    a call of only empty batches does no MSM at all.  (both used to return false with every verdict false.)"""
    for f in ("sign_message", "batch_verify_strict"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    rng = ecc.SplitMix64(909)
    keys = []
    for _ in range(5):
        sk = ecc.random_scalar(rng, ecc.R377)
        skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
        pkh = C.c_void_p()
        assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
        keys.append((skh, pkh))

    def sign(skh, m):
        s = C.c_void_p()
        assert sys_lib.sign_message(skh, m, C.c_int(len(m)), b"", C.c_int(0), CF, C22, C.byref(s))
        return s

    def batch(msg, n, spoil=None):
        sig5 = [sign(sk, msg) for sk, _ in keys]
        pks = (C.c_void_p * n)(*[keys[i % 5][1].value for i in range(n)])
        sgl = [sig5[i % 5].value for i in range(n)]
        if spoil is not None:
            sgl[spoil] = sign(keys[spoil % 5][0], b"something else").value
        sgs = (C.c_void_p * n)(*sgl)
        return (pks, sgs, msg, sig5), _BatchMessageFFI(_Buffer(msg, len(msg)), _Buffer(b"", 0), pks, n, sgs, n)

    h_big, big = batch(b"a batch of 1025 signers", 1025)
    h_bad, bad = batch(b"another large batch", 1100, spoil=1033)
    h_small, small = batch(b"a small batch beside them", 7)
    empty = _BatchMessageFFI(_Buffer(b"nobody signed this", 18), _Buffer(b"", 0), None, 0, None, 0)
    arr = (_BatchMessageFFI * 4)(big, empty, small, bad)
    res = (C.c_bool * 4)()
    assert not sys_lib.batch_verify_strict(arr, C.c_size_t(4), CF, C22, res)
    assert list(res) == [True, True, True, False]
    arr = (_BatchMessageFFI * 3)(big, empty, small)
    res = (C.c_bool * 3)()
    assert sys_lib.batch_verify_strict(arr, C.c_size_t(3), CF, C22, res) and list(res) == [True, True, True]
    only_empty = (_BatchMessageFFI * 2)(empty, empty)
    res = (C.c_bool * 2)()
    assert sys_lib.batch_verify_strict(only_empty, C.c_size_t(2), CF, C22, res) and list(res) == [True, True]


@pytest.mark.gpu
def test_batch_verify_signature_4097_pair_product(sys_lib, gpu):
    """BASELINE config 3, second shape: Signature::batch_verify over n = 4096 aggregates is ONE product of
    4097 pairs, e(sum sig_i, -g2) * prod e(H(m_i), apk_i) == 1 (crates/bls-crypto/src/bls/signature.rs:101-155).  4096 epochs with
    aggregate keys of 3 validators accept; one changed message, or one signature swapped for another epoch's, rejects."""
    for f in ("sign_message", "batch_verify_signature", "aggregate_public_keys", "aggregate_signatures"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    rng = ecc.SplitMix64(4097)
    n = 4096
    vals = []
    for _ in range(3):
        sk = ecc.random_scalar(rng, ecc.R377)
        skh = _deser(sys_lib, "deserialize_private_key", sk.to_bytes(32, "little"))
        pkh = C.c_void_p()
        assert sys_lib.private_key_to_public_key(skh, C.byref(pkh))
        vals.append((sk, skh, pkh))
    apk = C.c_void_p()
    assert sys_lib.aggregate_public_keys((C.c_void_p * 3)(*[v[2].value for v in vals]), C.c_int(3), C.byref(apk))
    # the aggregate signature of an epoch = (sk_1 + sk_2 + sk_3) H(m): signed once with the summed key (same group element)
    sk_sum = sum(v[0] for v in vals) % ecc.R377
    sum_h = _deser(sys_lib, "deserialize_private_key", sk_sum.to_bytes(32, "little"))
    msgs = [b"epoch %d of 4096" % i for i in range(n)]
    sigs = []
    for m in msgs:
        s = C.c_void_p()
        assert sys_lib.sign_message(sum_h, m, C.c_int(len(m)), b"x", C.c_int(1), CF, C22, C.byref(s))
        sigs.append(s)
    # the first epoch's aggregate built the long way, as a check of the shortcut
    parts = []
    for _, skh, _ in vals:
        s = C.c_void_p()
        assert sys_lib.sign_message(skh, msgs[0], C.c_int(len(msgs[0])), b"x", C.c_int(1), CF, C22, C.byref(s))
        parts.append(s)
    a0 = C.c_void_p()
    assert sys_lib.aggregate_signatures((C.c_void_p * 3)(*[p.value for p in parts]), C.c_int(3), C.byref(a0))
    assert _ser(sys_lib, "serialize_signature", a0) == _ser(sys_lib, "serialize_signature", sigs[0])

    def run(ms, sg):
        arr = (_MessageFFI * n)(*[_MessageFFI(_Buffer(ms[i], len(ms[i])), _Buffer(b"x", 1), apk.value, sg[i].value) for i in range(n)])
        ok = C.c_bool(False)
        assert sys_lib.batch_verify_signature(arr, C.c_size_t(n), CF, C22, C.byref(ok))
        return ok.value

    assert run(msgs, sigs)
    bad = list(msgs)
    bad[4095] = b"epoch 4095 of 4097"
    assert not run(bad, sigs)
    # swapping two signatures leaves the SUM of the signatures unchanged, so the product still holds (the check is on the aggregate,
    # as in the reference); replacing one by a foreign signature does not
    swapped = list(sigs)
    swapped[7], swapped[4000] = swapped[4000], swapped[7]
    assert run(msgs, swapped)
    foreign = list(sigs)
    foreign[2048] = sigs[2049]
    assert not run(msgs, foreign)


@pytest.mark.gpu
def test_batch_verify_strict_config3_scale_through_the_ffi(sys_lib, gpu):
    """BASELINE config 3 through the reference-named C ABI: 4096 batches x 256 signers in ONE batch_verify_strict call
    (crates/bls-snark-sys/src/signatures.rs:343-400), handles as they arrive from the wire, OS-RNG exponents; all batches accept,
    and with two batches carrying a foreign signature exactly those two are rejected."""
    for f in ("sign_message", "batch_verify_strict", "generate_private_key"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    NK, NS, m, nmsg = 16, 256, 4096, 8

    def roundtrip(h, ser, deser):
        return _deser(sys_lib, deser, _ser(sys_lib, ser, h))

    keys = []
    for _ in range(NK):
        sk, pk = C.c_void_p(), C.c_void_p()
        assert sys_lib.generate_private_key(C.byref(sk)) and sys_lib.private_key_to_public_key(sk, C.byref(pk))
        keys.append((sk, roundtrip(pk, "serialize_public_key", "deserialize_public_key")))
    per_msg = []
    for b in range(nmsg):
        msg = b"epoch-%06d" % b
        sg = []
        for sk, _ in keys:
            s = C.c_void_p()
            assert sys_lib.sign_message(sk, msg, C.c_int(len(msg)), b"", C.c_int(0), CF, C22, C.byref(s))
            sg.append(roundtrip(s, "serialize_signature", "deserialize_signature"))
        per_msg.append((msg, (C.c_void_p * NS)(*[keys[i % NK][1].value for i in range(NS)]), (C.c_void_p * NS)(*[sg[i % NK].value for i in range(NS)]), sg))
    arr = (_BatchMessageFFI * m)()
    for b in range(m):
        msg, pk_arr, sg_arr, _ = per_msg[b % nmsg]
        arr[b] = _BatchMessageFFI(_Buffer(msg, len(msg)), _Buffer(b"", 0), pk_arr, NS, sg_arr, NS)
    out = (C.c_bool * m)()
    assert sys_lib.batch_verify_strict(arr, C.c_size_t(m), CF, C22, out) and all(out)
    # two spoiled batches: signer 100's signature replaced by the one it gave for another message
    spoiled = {}
    for b in (17, 4001):
        msg, pk_arr, sg_arr, _ = per_msg[b % nmsg]
        other = per_msg[(b + 1) % nmsg][3]
        sl = [sg_arr[i] for i in range(NS)]
        sl[100] = other[100 % NK].value
        spoiled[b] = (C.c_void_p * NS)(*sl)
        arr[b] = _BatchMessageFFI(_Buffer(msg, len(msg)), _Buffer(b"", 0), pk_arr, NS, spoiled[b], NS)
    assert not sys_lib.batch_verify_strict(arr, C.c_size_t(m), CF, C22, out)
    assert [b for b in range(m) if not out[b]] == [17, 4001]


@pytest.mark.gpu
def test_batch_verify_strict_two_host_threads_overlap(sys_lib, gpu):
    """The reference's batch_verify_strict is re-entrant (crates/bls-snark-sys/src/signatures.rs:343: no shared state but the key cache).  Two
    host threads, each verifying its own 2048 batches x 256 signers (one spoiled batch each): the verdicts are those of the calls made one
    after the other, and the two calls overlap: the per-device lock covers the mirror phase only (csrc/seam_a.hip DevStage), the MSMs and
    pairing checks of the two calls run on pooled engines side by side.  What bounds the pair is the GPU, not a lock:
    a warm call of this size is 0.8 ms of host passes + 2.6 ms waiting for the message hashes + 13.9 ms of device chain (CELO_AMD_LOG=1,
    profiles/r5_strict_two_threads.txt) = 17.3 ms, i.e. 80 % GPU-busy, so two calls cannot finish in less than ~1.6-1.7 x one; measured
    1.72 x (29.8 ms against 34.6 ms for the two calls one after the other; 33.6 = 1.93 x with the call-long lock of rounds 3-4) on one box
    of the pool and 1.89 x (32.4 against 34.3 ms) on another, where the two calls' host passes compete for the same cores.  What is ASSERTED
    is the part that does not depend on the box: the verdicts under concurrency, and that the pair is not slower than the two calls back to
    back (best of five); the ratio is printed."""
    import threading
    import time
    for f in ("sign_message", "batch_verify_strict", "generate_private_key"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    NK, NS, m, nmsg = 16, 256, 2048, 8

    def roundtrip(h, ser, deser):
        return _deser(sys_lib, deser, _ser(sys_lib, ser, h))

    def workload(tag, bad):
        keys = []
        for _ in range(NK):
            sk, pk = C.c_void_p(), C.c_void_p()
            assert sys_lib.generate_private_key(C.byref(sk)) and sys_lib.private_key_to_public_key(sk, C.byref(pk))
            keys.append((sk, roundtrip(pk, "serialize_public_key", "deserialize_public_key")))
        per_msg = []
        for b in range(nmsg):
            msg = b"%s-epoch-%06d" % (tag, b)
            sg = []
            for sk, _ in keys:
                s_ = C.c_void_p()
                assert sys_lib.sign_message(sk, msg, C.c_int(len(msg)), b"", C.c_int(0), CF, C22, C.byref(s_))
                sg.append(roundtrip(s_, "serialize_signature", "deserialize_signature"))
            per_msg.append((msg, (C.c_void_p * NS)(*[keys[i % NK][1].value for i in range(NS)]), (C.c_void_p * NS)(*[sg[i % NK].value for i in range(NS)]), sg))
        arr = (_BatchMessageFFI * m)()
        keep = [per_msg]
        for b in range(m):
            msg, pk_arr, sg_arr, _ = per_msg[b % nmsg]
            arr[b] = _BatchMessageFFI(_Buffer(msg, len(msg)), _Buffer(b"", 0), pk_arr, NS, sg_arr, NS)
        msg, pk_arr, sg_arr, _ = per_msg[bad % nmsg]
        sl = [sg_arr[i] for i in range(NS)]
        sl[5] = per_msg[(bad + 1) % nmsg][3][5 % NK].value
        sp = (C.c_void_p * NS)(*sl)
        keep.append(sp)
        arr[bad] = _BatchMessageFFI(_Buffer(msg, len(msg)), _Buffer(b"", 0), pk_arr, NS, sp, NS)
        return arr, keep

    jobs = [workload(b"A", 77), workload(b"B", 1500)]
    outs = [(C.c_bool * m)(), (C.c_bool * m)()]
    rets = [None, None]

    def call(i):
        rets[i] = sys_lib.batch_verify_strict(jobs[i][0], C.c_size_t(m), CF, C22, outs[i])

    for i in range(2):                       # warm: engines, arenas, mirrors; and the sequential verdicts
        call(i)
        assert rets[i] is False and [b for b in range(m) if not outs[i][b]] == [(77, 1500)[i]]
    one = []
    for _ in range(3):
        t0 = time.perf_counter(); call(0); one.append(time.perf_counter() - t0)
    both = []
    for _ in range(5):
        th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        both.append(time.perf_counter() - t0)
        for i in range(2):
            assert rets[i] is False and [b for b in range(m) if not outs[i][b]] == [(77, 1500)[i]]
    print("one call %.2f ms, two concurrent calls %.2f ms = %.2f x" % (min(one) * 1e3, min(both) * 1e3, min(both) / min(one)))
    assert min(both) < 2 * min(one), (one, both)


@pytest.mark.gpu
def test_batch_verify_strict_device_mirror_follows_the_handles(sys_lib, gpu):
    """batch_verify_strict keeps the points of the handles it has seen in HBM, by arena slot (csrc/seam_a.hip HandleArena / Mirror), and
    ships slot numbers.  What must hold whatever the mirror holds: a repeated call over the same handles gives the same verdicts; a
    slot that is destroyed and handed out again carries its NEW point (the reused slot's stale row must not be paired with the new
    handle); a destroyed handle in a list makes the call fail instead of verifying against a stale row; handles that are not affine
    (aggregated keys and signatures, Z != 1) go through the same path.  Large enough (>= 8192 signers) for the threaded host pass."""
    for f in ("sign_message", "batch_verify_strict", "generate_private_key", "aggregate_public_keys", "aggregate_signatures"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    NK, NS, m = 8, 128, 80

    def fresh_key():
        sk, pk = C.c_void_p(), C.c_void_p()
        assert sys_lib.generate_private_key(C.byref(sk)) and sys_lib.private_key_to_public_key(sk, C.byref(pk))
        return sk, pk

    def sign(sk, msg):
        s = C.c_void_p()
        assert sys_lib.sign_message(sk, msg, C.c_int(len(msg)), b"", C.c_int(0), CF, C22, C.byref(s))
        return s

    keys = [fresh_key() for _ in range(NK)]                  # fresh public keys: Jacobian with Z != 1 (normalised on upload)
    msgs = [b"mirror-%03d" % b for b in range(m)]
    sigs = [[sign(sk, msg) for sk, _ in keys] for msg in msgs]

    def call(pk_of, sig_of):
        keep, arr = [], (_BatchMessageFFI * m)()
        for b in range(m):
            pks = (C.c_void_p * NS)(*[pk_of(b, i).value for i in range(NS)])
            sgs = (C.c_void_p * NS)(*[sig_of(b, i).value for i in range(NS)])
            keep.append((pks, sgs))
            arr[b] = _BatchMessageFFI(_Buffer(msgs[b], len(msgs[b])), _Buffer(b"", 0), pks, NS, sgs, NS)
        out = (C.c_bool * m)()
        rc = sys_lib.batch_verify_strict(arr, C.c_size_t(m), CF, C22, out)
        return rc, list(out)

    plain = (lambda b, i: keys[i % NK][1], lambda b, i: sigs[b][i % NK])
    assert call(*plain) == (True, [True] * m)
    assert call(*plain) == (True, [True] * m)                # second call: nothing to upload, slot numbers only
    # destroy key 3's public key and signer 3's signature of batch 5; the arenas hand the slots out again (LIFO free list)
    old_pk, old_sig = keys[3][1], sigs[5][3]
    assert sys_lib.destroy_public_key(old_pk) and sys_lib.destroy_signature(old_sig)
    sk_new, pk_new = fresh_key()
    assert pk_new.value == old_pk.value, "the arena reuses the released slot (this test relies on it)"
    sig_new = sign(sk_new, msgs[5])
    assert sig_new.value == old_sig.value
    # the reused key slot now holds ANOTHER key: batches signed by the old key 3 must fail everywhere, except batch 5 whose signer-3
    # signature slot now holds the new key's signature of that message
    keys[3] = (keys[3][0], pk_new)
    sigs[5][3] = sig_new
    rc, out = call(*plain)
    assert not rc and out == [b == 5 for b in range(m)]
    # sign everything for the new key: all batches verify again
    for b in range(m):
        if b != 5:
            sigs[b][3] = sign(sk_new, msgs[b])
    assert call(*plain) == (True, [True] * m)
    # a destroyed handle in the lists: the call fails, every verdict false - and the mirror is still right afterwards
    victim = sigs[7][0]
    assert sys_lib.destroy_signature(victim)
    rc, out = call(*plain)
    assert not rc and out == [False] * m
    sigs[7][0] = sign(keys[0][0], msgs[7])
    assert call(*plain) == (True, [True] * m)
    # aggregated handles (sums: Z != 1): batch b = one aggregate key with one aggregate signature, NS times over
    agg_pk = C.c_void_p()
    arr_pk = (C.c_void_p * NK)(*[k[1].value for k in keys])
    assert sys_lib.aggregate_public_keys(arr_pk, C.c_int(NK), C.byref(agg_pk))
    agg_sigs = []
    for b in range(m):
        a = C.c_void_p()
        assert sys_lib.aggregate_signatures((C.c_void_p * NK)(*[s.value for s in sigs[b]]), C.c_int(NK), C.byref(a))
        agg_sigs.append(a)
    assert call(lambda b, i: agg_pk, lambda b, i: agg_sigs[b]) == (True, [True] * m)
    assert call(lambda b, i: agg_pk, lambda b, i: agg_sigs[(b + 1) % m]) == (False, [False] * m)


def test_handle_arena_allocation_and_reuse(sys_lib, golden):
    """Handles live in chunked arenas (csrc/seam_a.hip HandleArena): addresses are stable, a destroyed handle's slot is handed out again
    (LIFO), contents never leak from one tenant of a slot to the next, and more handles than one chunk holds (2^14) keep their
    contents - host only, through the reference-named symbols.  Concurrent create / destroy from several threads ends consistent."""
    import threading
    pts = [bytes.fromhex(h) for h in list(golden["hash_to_g1_non_compat"])[:6]] if "hash_to_g1_non_compat" in golden else None
    if not pts:
        rng = ecc.SplitMix64(4242)
        pts = [ecc.ser_point(ecc.E1_377, ecc.E1_377.mul(ecc.G1_377, rng.next() | 1)) for _ in range(6)]
    base = [_deser(sys_lib, "deserialize_signature", p) for p in pts]
    assert all(base)
    sys_lib.aggregate_signatures.restype = C.c_bool

    def clone(h):
        o = C.c_void_p()
        assert sys_lib.aggregate_signatures((C.c_void_p * 1)(h), C.c_int(1), C.byref(o))
        return o

    n = (1 << 14) + 4000                                                  # crosses a chunk boundary
    many = [clone(base[i % 6]) for i in range(n)]
    assert len({h.value for h in many}) == n
    for i in (0, 1, (1 << 14) - 1, 1 << 14, n - 1, 7777):
        assert _ser(sys_lib, "serialize_signature", many[i]) == pts[i % 6]
    # LIFO reuse, with the new tenant's contents
    a = many[100].value
    assert sys_lib.destroy_signature(many[100])
    many[100] = clone(base[5])
    assert many[100].value == a and _ser(sys_lib, "serialize_signature", many[100]) == pts[5]
    assert _ser(sys_lib, "serialize_signature", many[99]) == pts[99 % 6] and _ser(sys_lib, "serialize_signature", many[101]) == pts[101 % 6]
    for h in many:
        assert sys_lib.destroy_signature(h)
    again = [clone(base[(i + 1) % 6]) for i in range(n)]
    assert {h.value for h in again} == {h.value for h in many}          # the same slots, no growth
    for i in (0, 5000, n - 1):
        assert _ser(sys_lib, "serialize_signature", again[i]) == pts[(i + 1) % 6]
    for h in again:
        assert sys_lib.destroy_signature(h)
    assert not sys_lib.destroy_signature(again[0])                       # destroyed twice: refused, the slot enters the free list once
    # four threads creating and destroying at once: every handle reads back what its creator put in
    errs = []

    def worker(t):
        try:
            mine = []
            for r in range(3000):
                h = clone(base[(t + r) % 6])
                mine.append((h, (t + r) % 6))
                if r % 3 == 2:
                    hh, k = mine.pop(0)
                    if _ser(sys_lib, "serialize_signature", hh) != pts[k]:
                        errs.append((t, r))
                    assert sys_lib.destroy_signature(hh)
            for hh, k in mine:
                if _ser(sys_lib, "serialize_signature", hh) != pts[k]:
                    errs.append((t, -1))
                assert sys_lib.destroy_signature(hh)
        except Exception as e:      # noqa: BLE001
            errs.append((t, repr(e)))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:3]


@pytest.mark.gpu
def test_batch_verify_strict_mirror_randomised_lifecycle(sys_lib, gpu):
    """Six rounds of create / destroy / re-create of key and signature handles between batch_verify_strict calls of ~9000 signers (the
    threaded host pass), batches drawn at random over the live handles with some deliberately mismatched pairs: every verdict must be
    what the handles' CONTENTS say, whatever rows earlier calls left in the device mirror."""
    import random
    for f in ("sign_message", "batch_verify_strict", "generate_private_key"):
        getattr(sys_lib, f).restype = C.c_bool
    CF, C22 = C.c_bool(False), C.c_bool(False)
    rnd = random.Random(20260930)
    NK, NM = 10, 5
    msgs = [b"lifecycle-%d" % i for i in range(NM)]

    def fresh_key():
        sk, pk = C.c_void_p(), C.c_void_p()
        assert sys_lib.generate_private_key(C.byref(sk)) and sys_lib.private_key_to_public_key(sk, C.byref(pk))
        return sk, pk

    def sign(sk, msg):
        s = C.c_void_p()
        assert sys_lib.sign_message(sk, msg, C.c_int(len(msg)), b"", C.c_int(0), CF, C22, C.byref(s))
        return s

    keys = [fresh_key() for _ in range(NK)]
    gen = [0] * NK                                             # generation of key k's secret
    sigs = {(k, j): (sign(keys[k][0], msgs[j]), 0) for k in range(NK) for j in range(NM)}    # handle, generation it was signed under
    for rnd_no in range(6):
        # lifecycle: replace two keys (their old signatures become wrong for the new key until re-signed), re-sign some, re-create some handles as they are
        for k in rnd.sample(range(NK), 2):
            assert sys_lib.destroy_public_key(keys[k][1])
            keys[k] = fresh_key(); gen[k] += 1
        for (k, j) in rnd.sample(sorted(sigs), 12):
            h, g = sigs[(k, j)]
            assert sys_lib.destroy_signature(h)
            sigs[(k, j)] = (sign(keys[k][0], msgs[j]), gen[k])
        m = 300
        arr, keep, want = (_BatchMessageFFI * m)(), [], []
        for b in range(m):
            j = rnd.randrange(NM)
            n = rnd.randrange(1, 60)
            ks = [rnd.randrange(NK) for _ in range(n)]
            ok = True
            pl, sl = [], []
            for k in ks:
                pl.append(keys[k][1].value)
                if rnd.random() < 0.01:                        # a signature of another message: a deliberate mismatch
                    h, g = sigs[(k, (j + 1) % NM)]; ok = False
                else:
                    h, g = sigs[(k, j)]; ok = ok and g == gen[k]
                sl.append(h.value)
            pa, sa = (C.c_void_p * n)(*pl), (C.c_void_p * n)(*sl)
            keep.append((pa, sa))
            arr[b] = _BatchMessageFFI(_Buffer(msgs[j], len(msgs[j])), _Buffer(b"", 0), pa, n, sa, n)
            want.append(ok)
        out = (C.c_bool * m)()
        rc = sys_lib.batch_verify_strict(arr, C.c_size_t(m), CF, C22, out)
        assert list(out) == want, (rnd_no, [b for b in range(m) if out[b] != want[b]][:5])
        assert rc == all(want)


def test_crh_paths_agree(sys_lib):
    """The single-message composite CRH has two host implementations: the 64-bit-limb table (round 6, the default) and pedersen.h's 28-bit-limb
    source the GPU kernel shares (CELO_CRH_DEVICE_LIMBS=1).  Same 48 bytes on every length 0 ... 130 (chunk boundaries on every bit offset, the
    first two generator windows) and on a 1 500-byte message (windows 0 ... 43)."""
    import subprocess, sys, hashlib
    code = (
        "import ctypes as C, hashlib, sys\n"
        "lib = C.CDLL(sys.argv[1]); lib.hash_crh.restype = C.c_bool\n"
        "h = hashlib.sha256()\n"
        "for n in list(range(131)) + [1500]:\n"
        "    m = bytes((7 * i + n) & 255 for i in range(n)); out, k = C.c_void_p(), C.c_int()\n"
        "    assert lib.hash_crh(m, n, 96, C.byref(out), C.byref(k)) and k.value == 48\n"
        "    h.update(bytes(C.cast(out, C.POINTER(C.c_ubyte * 48)).contents))\n"
        "print(h.hexdigest())\n")
    from celo_bls_snark_rs_amd import ffi
    outs = []
    for env_extra in ({}, {"CELO_CRH_DEVICE_LIMBS": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code, ffi.LIB_PATH], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64
