#!/usr/bin/env python3
"""A/B of the batched MSM window sizes inside config 3 (chained Batch::verify, 4096 x 256): usage ab_cfg3.py [c_g1] [c_g2] (0 = automatic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from celo_bls_snark_rs_amd import ffi
cg1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cg2 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cx = bench.Ctx()
class A: pass
cx.args = A(); cx.args.batches = 4096; cx.args.signers = 256; cx.args.scaling = "weak"; cx.args.no_cpu_baseline = True
cx.world, cx.rank, cx.cfg = 1, 0, 3
torch.cuda.set_device(0)
ffi.init(0)
job = bench.BatchVerifyConfig(cx)
job.setup()
if cg1: ffi.set_window_bits("bls12_377_g1", cg1)
if cg2: ffi.set_window_bits("bls12_377_g2", cg2)
for _ in range(2): r = job.step()
ts = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = job.step(); ts.append((time.perf_counter() - t0) * 1e3); job.after_step()
assert r.tolist() == job.w["expect"].tolist()
d = np.median(np.array(job.dev_ms), axis=0)
print("c_g1=%d c_g2=%d  median %.2f ms  (G2 MSM %.2f acc %.2f, G1 MSM %.2f, pairings %.2f)  windows g1 %s g2 %s" % (cg1, cg2, np.median(ts), d[0], d[1], d[2], d[3],
      ffi.msm_timings("bls12_377_g1")["window_bits"], ffi.msm_timings("bls12_377_g2")["window_bits"]))
