import sys, time, threading, json
sys.path.insert(0, "/root/repo")
import numpy as np
from celo_bls_snark_rs_amd import ffi, synthetic as syn
ffi.init(0)
n = 1 << 20
jobs = []
for i in range(4):
    xy = syn.device_points("bls12_377_g1", n, 100 + i).cpu().numpy().view(np.uint64).reshape(n, 12).copy()
    sc = syn.uniform_scalars("bls12_377_g1", n, 200 + i)
    jobs.append((xy, sc))
out = {}
for T in (1, 2, 4):
    for name, k in (("plain", 0), ("pipelined", -1)):
        ffi.set_host_chunks(k)
        def run(i):
            for _ in range(6):
                ffi.msm("bls12_377_g1", jobs[i][0], None, jobs[i][1])
        for i in range(T): run(i)   # warm
        th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        out["%d threads %s ms per MSM" % (T, name)] = round((time.perf_counter() - t0) * 1e3 / (6 * T), 3)
ffi.set_host_chunks(-1)
print(json.dumps(out))
