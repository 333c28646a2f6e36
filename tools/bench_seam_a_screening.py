#!/usr/bin/env python3
"""End-to-end batch_verify_signature through the reference-named C ABI: n (aggregated key, message, aggregated signature)
triples = one (n+1)-pair product with ONE final exponentiation (Signature::batch_verify, signature.rs:101-155; the shape of
crates/bls-crypto/benches/batch_bls.rs "all epoch aggregate screening" with n = 300)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celo_bls_snark_rs_amd import ffi
lib = C.CDLL(ffi.LIB_PATH)
for f in ("init", "generate_private_key", "private_key_to_public_key", "sign_message", "batch_verify_signature", "aggregate_signatures"):
    getattr(lib, f).restype = C.c_bool
assert lib.init()


class Buffer(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_size_t)]


class MessageFFI(C.Structure):
    _fields_ = [("data", Buffer), ("extra", Buffer), ("public_key", C.c_void_p), ("sig", C.c_void_p)]


sk, pk = C.c_void_p(), C.c_void_p()
assert lib.generate_private_key(C.byref(sk)) and lib.private_key_to_public_key(sk, C.byref(pk))
res = {}
for n in [int(a) for a in sys.argv[1:]] or [300, 4096]:
    msgs, keep = [], []
    for e in range(n):
        m = b"epoch-%06d" % e
        s = C.c_void_p()
        assert lib.sign_message(sk, m, len(m), b"", 0, C.c_bool(False), C.c_bool(False), C.byref(s))
        keep.append((m, s))
        msgs.append(MessageFFI(Buffer(m, len(m)), Buffer(b"", 0), pk.value, s.value))
    arr = (MessageFFI * n)(*msgs)
    ok = C.c_bool(False)
    assert lib.batch_verify_signature(arr, C.c_size_t(n), C.c_bool(False), C.c_bool(False), C.byref(ok)) and ok.value
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); lib.batch_verify_signature(arr, C.c_size_t(n), C.c_bool(False), C.c_bool(False), C.byref(ok)); ts.append(time.perf_counter() - t0)
    res[n] = {"wall_ms": min(ts) * 1e3, "messages_per_s": n / min(ts)}
print(json.dumps(res))
