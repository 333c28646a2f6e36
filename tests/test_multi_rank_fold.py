"""CPU (-m "not gpu"): the N > 1 path of bench.py / SURVEY.md §8e with world_size 2 over gloo.

Each rank owns an index-range shard, produces its partial MSM result, the 144-byte Jacobian partials are exchanged with
ONE all_gather and every rank folds them with the product's host-side `celo_amd_sum_jacobian_*`.  There is no GPU here,
so the per-rank partial comes from the oracle; what is under test is the exchange + fold (product code) - for the index-range
partition (Folder) and for the window partition (WindowJoiner + msm_*_join_windows)."""
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
    # the WINDOW partition's exchange (bench.WindowJoiner: one all_gather of X || Y || ZZ || ZZZ + first bit, then msm_*_join_windows on
    # every rank).  Rank g's record here: the sum over ALL points of the scalar bits [b_g, b_g+1) - from the oracle, there is no GPU.
    bounds = [0, 130, 253] if world == 2 else [253 * g // world for g in range(world + 1)]
    b0, b1 = bounds[rank], bounds[rank + 1]
    sub = [(k >> b0) & ((1 << (b1 - b0)) - 1) for k in sc]
    Pg = ecc.E1_377.msm(pts, sub)
    rec = np.zeros(24, dtype=np.uint64)
    if Pg is not None:
        axy, _ = co.pack_g1_377([Pg])
        one = co.to_mont([1], ecc.Q377)[0]
        rec[:12] = axy.reshape(-1); rec[12:18] = one; rec[18:] = one
    total4 = bench.WindowJoiner(cx, "bls12_377_g1")(rec, b0)
    windows_ok = co.jac_to_affine(total4, "g1_377") == exp
    q.put((rank, got == exp, co.jac_to_affine(total2, "g1_377") == exp and co.jac_to_affine(total3, "g1_377") == exp and gathered_ok and windows_ok))
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


def test_bench_gpus_flag_is_not_decorative(monkeypatch):
    """bench.py --gpus N (the flag was parsed and ignored).  Without a launcher around it, N > 1 re-runs the command
    line as N ranks under torch.distributed.run on 127.0.0.1; under a launcher whose WORLD_SIZE differs from --gpus it refuses to
    run - so a line whose n_gpus differs from --gpus cannot be printed."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--scaling", "strong"])
    assert bench.launch_ranks(2) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--scaling", "strong"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.undo()
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")
