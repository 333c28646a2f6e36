"""CPU (-m "not gpu"): the chunk plan of the pipelined host-pointer MSM entry (csrc/runtime.h host_chunk_plan, exported as
celo_amd_msm_host_chunk_plan - pure host arithmetic, no device call).  The pipeline (csrc/msm.h run_device_windows' HostIn) relies on:
the chunks cover the n terms exactly once in order, no chunk is longer than the capacity cm that spaces the chunks' virtual indices on
the device, cm and every length but the last are multiples of 1024 (the digit rows and the sort's tiles assume it), nothing is empty,
the first chunk is the shortest when it is split (it is the one transfer nothing overlaps), and the count stays within the 80 the
engine's arrays hold.  tests/test_host_pipeline_gpu.py runs such plans on the device against the oracle."""
import numpy as np
import pytest


def _plan(n, k, h, t):
    from celo_bls_snark_rs_amd import ffi
    return ffi.host_chunk_plan(n, k, h, t)


def _check(n, k, h, t):
    got = _plan(n, k, h, t)
    assert got is not None, (n, k, h, t)
    cm, lens = got
    assert 1 <= len(lens) <= 80
    assert sum(lens) == n, (n, k, h, t, lens)
    assert all(0 < x <= cm for x in lens), (n, k, h, t, cm, lens)
    assert cm % 1024 == 0 and all(x % 1024 == 0 for x in lens[:-1]), (cm, lens)
    # (len(lens) * cm, the virtual index space, may pass 2^32 / windows for n near 2^30: run_host_windows checks it in 64 bits and
    # falls back to the plain form)
    assert len(lens) >= 2 and len(lens) <= k + h + t
    if h and lens[0] < cm:                                              # a split head: the pieces of the first cm points, smallest first
        j, acc = 0, 0
        while acc < cm:
            acc += lens[j]; j += 1
        assert acc == cm and 2 <= j <= h + 1 and lens[:j] == sorted(lens[:j]), (lens, h, cm)
    return cm, lens


def test_default_plans_of_the_bench_sizes():
    # 2^20 G1 / G2 terms: 4 chunks, the first halved once;  BW6-761 2^21: 8 chunks
    cm, lens = _check(1 << 20, 4, 1, 0)
    assert cm == 1 << 18 and lens == [1 << 17, 1 << 17, 1 << 18, 1 << 18, 1 << 18]
    cm, lens = _check(1 << 21, 8, 1, 0)
    assert cm == 1 << 18 and lens == [1 << 17, 1 << 17] + [1 << 18] * 7
    cm, lens = _check(1 << 20, 4, 2, 1)
    assert lens == [1 << 16, 1 << 16, 1 << 17, 1 << 18, 1 << 18, 1 << 17, 1 << 17]


def test_plan_invariants_over_ragged_sizes():
    rng = np.random.default_rng(20260930)
    sizes = [(1 << 17), (1 << 17) + 1, (1 << 17) + 4321, (1 << 18) - 5, (1 << 18) + 1, (1 << 20) - 1, (1 << 20) + 1023, (1 << 24) + 77, (1 << 30) - 1]
    sizes += [int(x) for x in rng.integers(1 << 17, 1 << 26, size=300)]
    for n in sizes:
        for k in (2, 3, 4, 7, 8, 16, 64):
            if n < (k << 16):
                assert _plan(n, k, 1, 0) is None                          # the entry points lower the chunk count instead (chunks <= n >> 16)
                continue
            for h, t in ((0, 0), (1, 0), (2, 1), (3, 3), (8, 8)):
                _check(n, k, h, t)


def test_plan_refuses_what_the_entry_points_do_not_pipeline():
    assert _plan(1 << 20, 0, 0, 0) is None and _plan(1 << 20, 65, 0, 0) is None
    # one chunk (set by hand only): the whole job, cut by the head split alone
    assert _plan((1 << 20) + 5, 1, 0, 0) == (((1 << 20) + 1024), [(1 << 20) + 5])
    cm, lens = _plan(1 << 20, 1, 2, 3)
    assert cm == 1 << 20 and lens == [1 << 18, 1 << 18, 1 << 19]
    assert _plan(1 << 30, 4, 0, 0) is None and _plan((1 << 17) - 1, 2, 0, 0) is None
    assert _plan(1 << 20, 4, 9, 0) is None and _plan(1 << 20, 4, 0, -1) is None
