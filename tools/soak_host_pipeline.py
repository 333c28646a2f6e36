#!/usr/bin/env python3
"""Randomised soak of the pipelined host-pointer MSM entry against the resident entry on the same terms (both from this library; the oracle
comparisons are tests/test_host_pipeline_gpu.py): random sizes, chunk counts, head / tail splits, infinity flags, scalar shapes (uniform,
witness-like, all equal), pageable and page-locked buffers.  Prints one JSON line; exit 1 on the first mismatch.
  python tools/soak_host_pipeline.py [--cases 200] [--seed 1] [--group bls12_377_g1]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--group", default="bls12_377_g1")
    ap.add_argument("--max-log-n", type=int, default=19)
    a = ap.parse_args()
    from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec
    ffi.init(0)
    rng = np.random.default_rng(a.seed)
    A, S, _ = ffi.GROUP_SHAPE[a.group]
    p = codec.Q377 if a.group.startswith("bls12_377") else codec.Q761
    ext = 2 if a.group == "bls12_377_g2" else 1
    nmax = 1 << a.max_log_n
    pool = syn.device_points(a.group, nmax, 0x50AC0000 + a.seed).cpu().numpy().view(np.uint64).reshape(nmax, A)
    done = 0
    for case in range(a.cases):
        n = int(rng.integers(1 << 17, nmax + 1))
        if case % 7 == 0:
            n = (n >> 10) << 10                                   # exact multiples of the chunk granule too
        k = int(rng.integers(1, max(3, min(12, n >> 16) + 1)))
        hs, ts = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        start = int(rng.integers(0, nmax - n + 1))
        xy = pool[start:start + n].copy()
        sc = syn.uniform_scalars(a.group, n, int(rng.integers(1, 1 << 30)))
        shape = case % 4
        if shape == 1:                                            # witness-like: many zeros and ones
            kind = rng.integers(0, 10, size=n)
            sc[kind < 4] = 0
            one = np.zeros(S, dtype=np.uint64); one[0] = 1
            sc[(kind >= 4) & (kind < 7)] = one
        elif shape == 2:                                          # one scalar for all: one bucket per window holds everything
            sc[:] = sc[0]
        inf = None
        if case % 3 == 0:
            inf = (rng.integers(0, 50, size=n) == 0).astype(np.uint8)
        d_xy = torch.from_numpy(xy.view(np.int64).reshape(-1)).cuda()
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        d_inf = torch.from_numpy(inf).cuda() if inf is not None else None
        want = codec.jacobian_to_affine(ffi.msm_dev(a.group, d_xy.data_ptr(), d_inf.data_ptr() if d_inf is not None else 0, d_sc.data_ptr(), n), p, ext)
        ffi.set_host_chunks(k, hs, ts)
        if case % 5 == 4:
            pb, ps = ffi.PinnedArray(xy.shape, np.uint64), ffi.PinnedArray(sc.shape, np.uint64)
            pb.a[...] = xy; ps.a[...] = sc
            got = ffi.msm(a.group, pb.a, inf, ps.a)
            pb.close(); ps.close()
        else:
            got = ffi.msm(a.group, xy, inf, sc)
        ffi.set_host_chunks(-1)
        if codec.jacobian_to_affine(got, p, ext) != want:
            print(json.dumps({"ok": False, "case": case, "n": n, "chunks": k, "head_split": hs, "tail_split": ts, "shape": shape, "flags": inf is not None, "seed": a.seed}))
            sys.exit(1)
        done += 1
    print(json.dumps({"ok": True, "group": a.group, "cases": done, "seed": a.seed, "sizes": "2^17 .. 2^%d" % a.max_log_n}))


if __name__ == "__main__":
    main()
