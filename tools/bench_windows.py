#!/usr/bin/env python3
"""Window partition of one MSM (msm_*_window_shard_dev + msm_*_join_windows): per-shard call time of every shard of a `world`-way
partition run ALONE on this device (what one rank of a one-process-per-GPU job spends), the join, and the in-process form with the
device listed `world` times.  usage: bench_windows.py [g1|g2|bw6] [log_n ...] [--world 8] [--subgroup]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn
ffi.init(0)
args = sys.argv[1:]
GROUP = "bls12_377_g1"
if args and args[0] in ("g1", "g2", "bw6"):
    GROUP = {"g1": "bls12_377_g1", "g2": "bls12_377_g2", "bw6": "bw6_761_g1"}[args[0]]
    args = args[1:]
sub = "--subgroup" in args
args = [a for a in args if a != "--subgroup"]
worlds = [8]
if "--world" in args:
    i = args.index("--world"); worlds = [int(x) for x in args[i + 1].split(",")]; args = args[:i] + args[i + 2:]
for logn in [int(a) for a in args] or [20]:
    n = 1 << logn
    b = syn.device_points(GROUP, n, 5)
    sc = syn.uniform_scalars(GROUP, n, 6)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    for _ in range(2):
        ffi.msm_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, subgroup=sub)
    t = []
    for _ in range(7):
        t0 = time.perf_counter(); ffi.msm_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, subgroup=sub); t.append((time.perf_counter() - t0) * 1e3)
    whole = float(np.median(t))
    row = {"log_n": logn, "group": GROUP, "subgroup": sub, "single_call_ms": round(whole, 3)}
    for world in worlds:
        shards = []
        recs, bits = [], []
        for r in range(world):
            for _ in range(2):
                ffi.msm_window_shard_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, r, world, subgroup=sub)
            t = []
            for _ in range(7):
                t0 = time.perf_counter(); rec, bit = ffi.msm_window_shard_dev(GROUP, b.data_ptr(), 0, d.data_ptr(), n, r, world, subgroup=sub); t.append((time.perf_counter() - t0) * 1e3)
            tm = ffi.msm_timings(GROUP)
            shards.append({"shard": r, "call_ms": round(float(np.median(t)), 3), "dev_ms": round(tm["total_ms"], 3), "conv": round(tm["convert_ms"], 3), "sort": round(tm["sort_ms"], 3),
                           "acc": round(tm["accumulate_ms"], 3), "red": round(tm["reduce_ms"], 3), "windows": tm["windows"], "bit_lo": bit})
            recs.append(rec); bits.append(bit)
        t = []
        for _ in range(7):
            t0 = time.perf_counter(); ffi.join_windows(GROUP, np.stack(recs), bits); t.append((time.perf_counter() - t0) * 1e3)
        join = float(np.median(t))
        worst = max(s["call_ms"] for s in shards)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); ffi.msm_multi_windows_dev(GROUP, [0] * world, [b.data_ptr()] * world, None, [d.data_ptr()] * world, n, subgroup=sub); t.append((time.perf_counter() - t0) * 1e3)
        row["world_%d" % world] = {"worst_shard_call_ms": worst, "join_ms": round(join, 3), "strong_scaling_bound": round(whole / (worst + join), 2),
                                   "in_process_same_device_ms": round(float(np.median(t)), 3), "shards": shards}
    print(json.dumps(row), flush=True)
