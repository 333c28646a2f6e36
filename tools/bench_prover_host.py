#!/usr/bin/env python3
"""groth16_prove_bw6_761 on HOST queries (the non-key form: every MSM takes the host-pointer entry that flags identity rows from the
bases), pipelined against plain:  python tools/bench_prover_host.py [log_rows=20]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec

ffi.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_inputs, n_aux = 3, (1 << logn) - 3
n_assign = n_inputs + n_aux
nh = (1 << logn) - 1
pts = lambda g, k, s: syn.device_points(g, k, s).cpu().numpy().view(np.uint64).reshape(k, 24)
a_q, b_q = pts("bw6_761_g1", n_assign + 1, 1), pts("bw6_761_g2", n_assign + 1, 2)
l_q, h_q = pts("bw6_761_g1", n_aux, 3), pts("bw6_761_g1", nh, 4)
alpha, beta = syn.generator_limbs("bw6_761_g1"), syn.generator_limbs("bw6_761_g2")
asg = syn.witness_like_scalars("bw6_761_g1", n_assign, 5)
h = syn.uniform_scalars("bw6_761_g1", nh, 6)
out = {"rows_per_query": 1 << logn}
ref = None
for name, k in (("plain", 0), ("pipelined", -1)):
    ffi.set_host_chunks(k)
    r = ffi.groth16_prove(a_q, b_q, h_q, l_q, alpha, beta, asg, n_aux, h)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); r = ffi.groth16_prove(a_q, b_q, h_q, l_q, alpha, beta, asg, n_aux, h); ts.append((time.perf_counter() - t0) * 1e3)
    aff = [codec.jacobian_to_affine(x, codec.Q761, 1) for x in r]
    if ref is None: ref = aff
    out[name + "_ms"] = float(np.median(ts)); out[name + "_equal"] = aff == ref
ffi.set_host_chunks(-1)
print(json.dumps(out))
