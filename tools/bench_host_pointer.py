#!/usr/bin/env python3
"""Sweep of the host-pointer MSM entry points (msm_<group> on pageable numpy buffers) over the number of index chunks of the pipelined
transfer (csrc/msm.h HostIn; celo_amd_msm_set_host_chunks), beside the resident entry point on the same inputs.
  python tools/bench_host_pointer.py [--group bls12_377_g1] [--log-n 20] [--chunks 0,2,4,8] [--reps 10] > profiles/r5_host_pointer_<group>.json"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", default="bls12_377_g1")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--chunks", default="0,2,3,4,6,8,12,16")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tail-split", default="", help="comma list: how often the last chunk is cut in halves (default: the library's)")
    ap.add_argument("--head-split", default="", help="comma list: how often the first chunk is cut in halves (default: the library's)")
    a = ap.parse_args()
    from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec
    ffi.init(0)
    n = 1 << a.log_n
    bases = syn.device_points(a.group, n, 0x5EED0002)
    sc = syn.uniform_scalars(a.group, n, 0x5EED0001)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    A = ffi.GROUP_SHAPE[a.group][0]
    h_bases = bases.cpu().numpy().view(np.uint64).reshape(n, A).copy()
    h_sc = np.ascontiguousarray(sc).copy()
    p = codec.Q377 if a.group.startswith("bls12_377") else codec.Q761
    ext = 2 if a.group == "bls12_377_g2" else 1
    for _ in range(3):
        ref = ffi.msm_dev(a.group, bases.data_ptr(), 0, d_sc.data_ptr(), n)
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ref = ffi.msm_dev(a.group, bases.data_ptr(), 0, d_sc.data_ptr(), n)
        ts.append((time.perf_counter() - t0) * 1e3)
    res_ms = float(np.median(ts))
    want = codec.jacobian_to_affine(ref, p, ext)
    out = {"group": a.group, "log_n": a.log_n, "resident_wall_ms": res_ms, "resident_kernel_ms": ffi.msm_timings(a.group), "bytes": h_bases.nbytes + h_sc.nbytes, "chunks": {}}
    splits = [int(x) for x in a.head_split.split(",")] if a.head_split else [None]
    tails = [int(x) for x in a.tail_split.split(",")] if a.tail_split else [None]
    for k, sp, tl in [(int(x), sp, tl) for x in a.chunks.split(",") for sp in splits for tl in tails]:
        ffi.set_host_chunks(k, sp, tl)
        for _ in range(2):
            o = ffi.msm(a.group, h_bases, None, h_sc)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            o = ffi.msm(a.group, h_bases, None, h_sc)
            ts.append((time.perf_counter() - t0) * 1e3)
        tm = ffi.msm_timings(a.group)
        # the same call on buffers the runtime has never seen (a caller that builds new Vecs per MSM): copies made outside the timing
        fr = []
        for _ in range(a.reps):
            fb, fs = h_bases.copy(), h_sc.copy()
            t0 = time.perf_counter()
            o2 = ffi.msm(a.group, fb, None, fs)
            fr.append((time.perf_counter() - t0) * 1e3)
            del fb, fs
        # ... and on page-locked buffers from celo_amd_host_alloc
        pb, ps = ffi.PinnedArray(h_bases.shape, np.uint64), ffi.PinnedArray(h_sc.shape, np.uint64)
        pb.a[...] = h_bases; ps.a[...] = h_sc
        ffi.msm(a.group, pb.a, None, ps.a)
        pn = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            o3 = ffi.msm(a.group, pb.a, None, ps.a)
            pn.append((time.perf_counter() - t0) * 1e3)
        pb.close(); ps.close()
        ok = codec.jacobian_to_affine(o, p, ext) == want and codec.jacobian_to_affine(o2, p, ext) == want and codec.jacobian_to_affine(o3, p, ext) == want
        out["chunks"][str(k) if sp is None and tl is None else "%d/%s/%s" % (k, sp, tl)] = {"wall_ms": float(np.median(ts)), "min_ms": float(np.min(ts)), "ratio_to_resident": float(np.median(ts)) / res_ms, "parity": ok,
                                 "fresh_buffers_wall_ms": float(np.median(fr)), "fresh_buffers_min_ms": float(np.min(fr)),
                                 "pinned_buffers_wall_ms": float(np.median(pn)), "pinned_buffers_min_ms": float(np.min(pn)),
                                 "kernel_ms": {q: round(tm[q], 3) for q in ("convert_ms", "sort_ms", "accumulate_ms", "reduce_ms", "total_ms")}}
        if not ok:
            raise SystemExit("PARITY FAILURE at chunks=%d" % k)
    ffi.set_host_chunks(-1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
