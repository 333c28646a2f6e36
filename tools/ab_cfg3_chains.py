#!/usr/bin/env python3
"""Config 3 (4096 x 256) through batch_verify_bls12_377_dev as ONE call and as K concurrent calls over K contiguous parts of the batches
(K host threads: each call leases its own engines and streams) - does cutting the batches into chains hide the latency-bound tails of one
chain under the accumulation of the next?  usage: ab_cfg3_chains.py [m=4096] [n=256] [K ...]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn
ffi.init(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Ks = [int(a) for a in sys.argv[3:]] or [1, 2, 3, 4]
w = syn.valid_batches(m, n, 0x5EED0300, [7, m // 2 + 3])
ex = syn.batch_exponents(m * n, 0x5EED0301)
d_ex = torch.from_numpy(ex.view(np.int64)).cuda()
ng2 = syn.neg_g2_limbs()
res = {}
for K in Ks:
    cuts = [m * k // K for k in range(K + 1)]
    outs = [None] * K

    def part(k):
        lo, hi = cuts[k], cuts[k + 1]
        offs = (w["offsets"][lo:hi + 1] - w["offsets"][lo]).astype(np.uint32)
        p0 = int(w["offsets"][lo])
        outs[k] = ffi.batch_verify_dev(w["pk"].data_ptr() + p0 * 192, w["sig"].data_ptr() + p0 * 96, d_ex.data_ptr() + p0 * 32, offs, w["hash"].data_ptr() + lo * 96, ng2)

    def run():
        th = [threading.Thread(target=part, args=(k,)) for k in range(K)]
        for t in th: t.start()
        for t in th: t.join()
    for _ in range(3):
        run()
    assert np.concatenate(outs).tolist() == w["expect"].tolist()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    res["chains_%d" % K] = {"median_ms": round(float(np.median(ts)), 3), "min_ms": round(min(ts), 3)}
    print(K, res["chains_%d" % K], flush=True)
print(json.dumps(res))
