#!/usr/bin/env python3
"""Chained Batch::verify (config-3 shape) through the HOST-buffer entry batch_verify_bls12_377 against the resident
batch_verify_bls12_377_dev on the same batches:  python tools/bench_batch_host.py [batches=4096] [signers=256]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn

ffi.init(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = syn.valid_batches(m, k, 0x5EED0300, [7, m // 2])
tot = m * k
ex = syn.batch_exponents(tot, 0x5EED0301)
d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
ng2 = syn.neg_g2_limbs()
def med(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r
t_dev, ok_dev = med(lambda: ffi.batch_verify_dev(w["pk"].data_ptr(), w["sig"].data_ptr(), d_ex.data_ptr(), w["offsets"], w["hash"].data_ptr(), ng2))
h_pk = w["pk"].cpu().numpy().view(np.uint64).copy(); h_sig = w["sig"].cpu().numpy().view(np.uint64).copy(); h_hash = w["hash"].cpu().numpy().view(np.uint64).copy()
t_host, ok_host = med(lambda: ffi.batch_verify(h_pk, h_sig, ex, w["offsets"], h_hash, ng2))
nbytes = h_pk.nbytes + h_sig.nbytes + ex.nbytes
print(json.dumps({"batches": m, "signers": k, "resident_ms": t_dev, "host_buffers_ms": t_host, "ratio": t_host / t_dev, "bytes": nbytes,
                  "h2d_at_56GBps_ms": nbytes / 56e9 * 1e3, "verdicts_equal": ok_dev.tolist() == ok_host.tolist(), "rejected": int((ok_dev == 0).sum())}))
