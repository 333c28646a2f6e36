#!/usr/bin/env python3
"""Latency of the single-call paths through the reference-named C ABI (Seam A) on one GPU:
verify_signature (one 2-pair BLS12-377 product, direct and composite hashers) and `verify` (Groth16 over BW6-761,
the reference's own FFI vector).  Prints one JSON object."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celo_bls_snark_rs_amd import ffi
lib = C.CDLL(ffi.LIB_PATH)
for f in ("init", "generate_private_key", "private_key_to_public_key", "sign_message", "verify_signature", "verify"):
    getattr(lib, f).restype = C.c_bool
t0 = time.perf_counter(); assert lib.init(); t_init = time.perf_counter() - t0
sk, pk, sig = C.c_void_p(), C.c_void_p(), C.c_void_p()
assert lib.generate_private_key(C.byref(sk)) and lib.private_key_to_public_key(sk, C.byref(pk))
res = {"init_s": t_init}
msg, extra = b"hello", b"extra"
for name, comp, cip in (("direct", False, False), ("composite", True, False), ("composite_cip22", True, True)):
    assert lib.sign_message(sk, msg, 5, extra, 5, C.c_bool(comp), C.c_bool(cip), C.byref(sig))
    ok = C.c_bool(False)
    lib.verify_signature(pk, msg, 5, extra, 5, sig, C.c_bool(comp), C.c_bool(cip), C.byref(ok)); assert ok.value
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); lib.verify_signature(pk, msg, 5, extra, 5, sig, C.c_bool(comp), C.c_bool(cip), C.byref(ok)); ts.append(time.perf_counter() - t0)
    res["verify_signature_%s_ms" % name] = min(ts) * 1e3


class EB(C.Structure):
    _fields_ = [("index", C.c_uint16), ("round", C.c_uint8), ("epoch_entropy", C.c_char_p), ("parent_entropy", C.c_char_p),
                ("pubkeys", C.c_char_p), ("pubkeys_num", C.c_size_t), ("maximum_non_signers", C.c_uint32), ("maximum_validators", C.c_size_t)]


g = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))["groth16_bw6_761"]
vk, proof = bytes.fromhex(g["vk"]), bytes.fromhex(g["proof"])
blk = lambda d, p, e, pe: EB(d["index"], d["round"], bytes.fromhex(g[e]), bytes.fromhex(g[pe]), bytes.fromhex(g[p]), d["pubkeys_num"], d["maximum_non_signers"], d["maximum_validators"])
first, last = blk(g["first"], "first_pubkeys", "first_epoch_entropy", "first_parent_entropy"), blk(g["last"], "last_pubkeys", "last_epoch_entropy", "last_parent_entropy")
lib.verify.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, EB, EB]
assert lib.verify(vk, len(vk), proof, len(proof), first, last)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assert lib.verify(vk, len(vk), proof, len(proof), first, last); ts.append(time.perf_counter() - t0)
res["groth16_verify_ms"] = min(ts) * 1e3
print(json.dumps(res))
