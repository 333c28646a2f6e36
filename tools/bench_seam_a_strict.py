#!/usr/bin/env python3
"""End-to-end batch_verify_strict through the reference-named C ABI (Seam A) at BASELINE config 3's shape: m batches x 256
signers, keys and signatures as deserialised handles (64 distinct key pairs per message, reused across batches), OS-RNG
exponents, direct or composite hasher.  Times the single FFI call: host packing + hashing + GPU MSMs + GPU pairings."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celo_bls_snark_rs_amd import ffi
lib = C.CDLL(ffi.LIB_PATH)
for f in ("init", "generate_private_key", "private_key_to_public_key", "sign_message", "batch_verify_strict", "serialize_public_key",
          "deserialize_public_key", "serialize_signature", "deserialize_signature", "free_vec"):
    getattr(lib, f).restype = C.c_bool
assert lib.init()


class Buffer(C.Structure):
    _fields_ = [("ptr", C.c_char_p), ("len", C.c_size_t)]


class BatchMessageFFI(C.Structure):
    _fields_ = [("data", Buffer), ("extra", Buffer), ("public_keys", C.POINTER(C.c_void_p)), ("public_keys_len", C.c_size_t),
                ("signatures", C.POINTER(C.c_void_p)), ("signatures_len", C.c_size_t)]


def roundtrip(handle, ser, deser):   # through the wire format, like keys and signatures arriving from the network
    out, n = C.c_void_p(), C.c_int()
    assert getattr(lib, ser)(handle, C.byref(out), C.byref(n))
    data = bytes(C.cast(out, C.POINTER(C.c_ubyte * n.value)).contents)
    lib.free_vec(out, n)
    h = C.c_void_p()
    assert getattr(lib, deser)(data, C.c_int(len(data)), C.byref(h))
    return h


NK, NS = 64, 256
sks = []
for _ in range(NK):
    sk, pk = C.c_void_p(), C.c_void_p()
    assert lib.generate_private_key(C.byref(sk)) and lib.private_key_to_public_key(sk, C.byref(pk))
    sks.append((sk, roundtrip(pk, "serialize_public_key", "deserialize_public_key")))
res = {}
for composite, cip22, name in ((False, False, "direct"), (True, True, "composite_cip22")):
    for m in [int(a) for a in sys.argv[1:]] or [256, 4096]:
        keep, batches = [], []
        nmsg = min(m, 32)                                     # distinct messages (signing on the host is the slow part of the setup)
        per_msg = []
        for b in range(nmsg):
            msg = b"epoch-%06d" % b
            sigs = []
            for sk, _ in sks:
                s = C.c_void_p()
                assert lib.sign_message(sk, msg, len(msg), b"", 0, C.c_bool(composite), C.c_bool(cip22), C.byref(s))
                sigs.append(roundtrip(s, "serialize_signature", "deserialize_signature"))
            pk_arr = (C.c_void_p * NS)(*[sks[i % NK][1].value for i in range(NS)])
            sg_arr = (C.c_void_p * NS)(*[sigs[i % NK].value for i in range(NS)])
            per_msg.append((msg, pk_arr, sg_arr))
        arr = (BatchMessageFFI * m)()
        for b in range(m):
            msg, pk_arr, sg_arr = per_msg[b % nmsg]
            arr[b] = BatchMessageFFI(Buffer(msg, len(msg)), Buffer(b"", 0), pk_arr, NS, sg_arr, NS)
        out = (C.c_bool * m)()
        assert lib.batch_verify_strict(arr, C.c_size_t(m), C.c_bool(composite), C.c_bool(cip22), out) and all(out)
        t0 = time.perf_counter()
        assert lib.batch_verify_strict(arr, C.c_size_t(m), C.c_bool(composite), C.c_bool(cip22), out)
        dt = time.perf_counter() - t0
        res["%s_m%d" % (name, m)] = {"wall_ms": dt * 1e3, "batches_per_s": m / dt, "signatures_per_s": m * NS / dt}
        if composite or os.environ.get("STRICT_DISTINCT", "1") == "0":
            continue
        # ---- the same call with a handle of its own behind EVERY position (m x 256 key handles + as many signature handles; copies made
        # with aggregate_*([h]) - a one-term sum).  first: nothing of them on the device yet (every row normalised / copied / uploaded);
        # steady: the same handles again (slot numbers only); fresh_sigs: the key handles stay, every signature handle is destroyed and
        # re-created between calls - what a caller sees whose validator keys persist while each block brings new signatures.
        for f in ("aggregate_public_keys", "aggregate_signatures", "destroy_signature"):
            getattr(lib, f).restype = C.c_bool

        def clone(h, fn):
            o = C.c_void_p()
            assert getattr(lib, fn)((C.c_void_p * 1)(h), C.c_int(1), C.byref(o))
            return o.value

        d_keep = []
        darr = (BatchMessageFFI * m)()

        def fill_sigs():
            for b in range(m):
                msg, pk_arr, sg_arr = per_msg[b % nmsg]
                d_keep[b][1][:] = [clone(sg_arr[i], "aggregate_signatures") for i in range(NS)]

        for b in range(m):
            msg, pk_arr, sg_arr = per_msg[b % nmsg]
            dp = (C.c_void_p * NS)(*[clone(pk_arr[i], "aggregate_public_keys") for i in range(NS)])
            ds = (C.c_void_p * NS)()
            d_keep.append((dp, ds))
            darr[b] = BatchMessageFFI(Buffer(msg, len(msg)), Buffer(b"", 0), dp, NS, ds, NS)
        fill_sigs()
        for label in ("first", "steady", "steady2", "fresh_sigs", "fresh_sigs2"):
            if label.startswith("fresh"):
                for b in range(m):
                    for i in range(NS):
                        assert lib.destroy_signature(C.c_void_p(d_keep[b][1][i]))
                fill_sigs()
            t0 = time.perf_counter()
            assert lib.batch_verify_strict(darr, C.c_size_t(m), C.c_bool(composite), C.c_bool(cip22), out) and all(out)
            dt = time.perf_counter() - t0
            res["%s_m%d_distinct_handles_%s" % (name, m, label)] = {"wall_ms": dt * 1e3, "batches_per_s": m / dt, "signatures_per_s": m * NS / dt}
print(json.dumps(res))
