#!/usr/bin/env python3
"""BASELINE.json config 1 ("bls-crypto simple_signature example: 64-validator aggregate sig verify", plumbing only): the flow of
crates/bls-crypto/examples/simple_signature.rs:11-67 through the bls-snark-sys C ABI this repository exports
(include/celo_bls_snark_sys.h) - generate N keys, every key signs the message with the COMPOSITE hasher (the example's
COMPOSITE_HASH_TO_G1), the keys and signatures are aggregated (in the example's nested way for N = 3: the third key and its
signature count twice), ONE verify_signature of the aggregate.  Signing is host code; the pairing check runs
on the GPU.

usage: python tools/simple_signature.py -m MESSAGE [-k KEYS (default 64; 3 = the reference example's shape)] [--direct] [--cip22]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celo_bls_snark_rs_amd import ffi   # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Show an example of a simple aggregated signature with random keys")
    ap.add_argument("-m", dest="message", required=True, help="Sets the message to sign")
    ap.add_argument("-k", dest="keys", type=int, default=64)
    ap.add_argument("--direct", action="store_true", help="direct (Blake2Xs) hasher instead of the composite one")
    ap.add_argument("--cip22", action="store_true")
    a = ap.parse_args()
    lib = C.CDLL(ffi.LIB_PATH)
    for f in ("init", "generate_private_key", "private_key_to_public_key", "sign_message", "aggregate_public_keys", "aggregate_signatures",
              "verify_signature", "serialize_private_key", "serialize_signature", "serialize_public_key", "free_vec"):
        getattr(lib, f).restype = C.c_bool
    if not lib.init():
        sys.exit("no gfx950 device: the verification path has no CPU fallback")
    comp, cip = C.c_bool(not a.direct), C.c_bool(a.cip22)
    msg = a.message.encode()
    print("matches: %s" % a.message)

    def hexof(fn, h):
        out, n = C.c_void_p(), C.c_int()
        assert getattr(lib, fn)(h, C.byref(out), C.byref(n))
        data = bytes(C.cast(out, C.POINTER(C.c_ubyte * n.value)).contents)
        lib.free_vec(out, n)
        return data.hex()
    sks, pks, sigs = [], [], []
    for i in range(a.keys):
        sk, pk, sg = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert lib.generate_private_key(C.byref(sk)) and lib.private_key_to_public_key(sk, C.byref(pk))
        assert lib.sign_message(sk, msg, C.c_int(len(msg)), b"", C.c_int(0), comp, cip, C.byref(sg))
        sks.append(sk); pks.append(pk); sigs.append(sg)
        if a.keys <= 4:
            print("sk%d: %s" % (i + 1, hexof("serialize_private_key", sk)))
            print("sig%d: %s" % (i + 1, hexof("serialize_signature", sg)))
    if a.keys == 3:       # the example's shape: apk = pk1 + pk2 + pk3 + pk3, asig = (sig1 + sig3) + (sig2 + sig3)
        pks.append(pks[2]); sigs.append(sigs[2])
    apk, asig = C.c_void_p(), C.c_void_p()
    assert lib.aggregate_public_keys((C.c_void_p * len(pks))(*[p.value for p in pks]), C.c_int(len(pks)), C.byref(apk))
    assert lib.aggregate_signatures((C.c_void_p * len(sigs))(*[s.value for s in sigs]), C.c_int(len(sigs)), C.byref(asig))
    print("apk: %s" % hexof("serialize_public_key", apk))
    print("asig: %s" % hexof("serialize_signature", asig))
    ok = C.c_bool(False)
    t0 = time.perf_counter()
    assert lib.verify_signature(apk, msg, C.c_int(len(msg)), b"", C.c_int(0), asig, comp, cip, C.byref(ok))
    dt = time.perf_counter() - t0
    if not ok.value:
        sys.exit("aggregated signature DID NOT verify")
    print("aggregated signature verified successfully (%d validators, verify_signature %.1f ms)" % (a.keys, dt * 1e3))


if __name__ == "__main__":
    main()
