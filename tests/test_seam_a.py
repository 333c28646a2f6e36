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
