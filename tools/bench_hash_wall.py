"""Wall time vs kernel time of the batched hash-to-G1 entry points at 4096 messages (host buffers in, host buffers out)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celo_bls_snark_rs_amd import ffi
ffi.init(0)
rng = np.random.default_rng(1)
n = 4096
raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
msgs = [raw[i].tobytes() for i in range(n)]
ex = [b"\x01\x02"] * n
lib = ffi.lib()
import ctypes as C
dom = np.frombuffer(b"ULforxof", dtype=np.uint8)
off = np.arange(0, 32 * (n + 1), 32, dtype=np.uint64); eoff = np.arange(0, 2 * (n + 1), 2, dtype=np.uint64)
md = raw.reshape(-1).copy(); ed = np.frombuffer(b"".join(ex), dtype=np.uint8)
xy = np.zeros((n, 12), dtype=np.uint64); att = np.zeros(n, dtype=np.uint8)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for name, call in (("direct", lambda: lib.hash_to_g1_direct_bls12_377(P(dom), P(md), P(off), P(ed), P(eoff), C.c_size_t(n), P(xy), P(att))),
                   ("composite_cip22", lambda: lib.hash_to_g1_composite_bls12_377(P(dom), P(md), P(off), P(ed), P(eoff), C.c_size_t(n), C.c_int(1), P(xy), P(att))),
                   ("composite", lambda: lib.hash_to_g1_composite_bls12_377(P(dom), P(md), P(off), P(ed), P(eoff), C.c_size_t(n), C.c_int(0), P(xy), P(att)))):
    call()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); rc = call(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "rc", rc, "wall_ms min %.2f" % min(ts), "last kernel_ms %.2f" % ffi.hash_last_ms())
