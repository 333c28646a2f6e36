"""CPU (-m "not gpu"): the N > 1 path of bench.py / SURVEY.md §8e with world_size 2 over gloo.

Each rank owns an index-range shard, produces its partial MSM result, the 144-byte Jacobian partials are exchanged with
ONE all_gather and every rank folds them with the product's host-side `celo_amd_sum_jacobian_*`.  There is no GPU here,
so the per-rank partial comes from the oracle; what is under test is the exchange + fold (product code)."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.py import ecc
    from oracle import cpu_oracle as co
    from celo_bls_snark_rs_amd import ffi
    from tests import helpers as H
    n = 64
    pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 900)
    sc = H.seeded_scalars(n, 901, ecc.R377)
    lo, hi = rank * n // world, (rank + 1) * n // world
    xy, inf = co.pack_g1_377(pts[lo:hi])
    part = co.msm("bls12_377_g1", xy, inf, H.scalars_np(sc[lo:hi], 4), threads=1)
    if rank == 1:
        part_t = torch.from_numpy(part.view(np.int64).copy())
    else:
        part_t = torch.from_numpy(part.view(np.int64).copy())
    bufs = [torch.empty(18, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bufs, part_t)
    parts = np.stack([b.numpy().view(np.uint64) for b in bufs])
    total = ffi.sum_jacobian("bls12_377_g1", parts)
    got = co.jac_to_affine(total, "g1_377")
    exp = ecc.E1_377.msm(pts, sc)
    # folding identity partials must be a no-op
    ident = np.zeros((1, 18), dtype=np.uint64)
    total2 = ffi.sum_jacobian("bls12_377_g1", np.concatenate([parts, ident]))
    # bench.py's own N > 1 plumbing (host-staged here): the Folder every MSM step goes through and the one-off gather of the
    # inputs to rank 0 behind the full-size parity check
    import bench
    cx = bench.Ctx()
    cx.world, cx.rank, cx.xdev = world, rank, "cpu"
    total3 = bench.Folder(cx, "bls12_377_g1", 18)(part)
    allxy = bench.gather_to_rank0(cx, torch.from_numpy(xy.view(np.int64).copy()))
    gathered_ok = True
    if rank == 0:
        full, _ = co.pack_g1_377(pts)
        gathered_ok = np.array_equal(allxy, full)
    q.put((rank, got == exp, co.jac_to_affine(total2, "g1_377") == exp and co.jac_to_affine(total3, "g1_377") == exp and gathered_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partial_sum_exchange():
    from celo_bls_snark_rs_amd import ffi
    if not os.path.exists(ffi.LIB_PATH):
        pytest.fail("library not built")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res)
