"""Multi-GPU through the C ABI.

* the WINDOW partition of one MSM (msm_*_multi_windows[_dev], msm_*_window_shard_dev + msm_*_join_windows): on the 1-GPU box the
  device is listed several times (that many engines on it) - every shard count, window layout (uniform, mixed widths, the GLV split's
  127-bit halves, more shards than windows) against the oracle;
* the join alone runs without a device (host arithmetic): CPU test against the big-integer group law;
* the tests that need TWO OR MORE devices skip on a 1-GPU box and, on a multi-GPU node, run both partitions over DISTINCT devices
  and `bench.py --gpus 2` over RCCL (backend nccl) with its full-size parity check.
The reference's callers are one process (crates/bls-snark-sys/src/signatures.rs:343, crates/epoch-snark/src/api/prover.rs:78)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

from oracle.py import ecc
from oracle import cpu_oracle as co
import helpers as H
import torch  # noqa: F401  (before the library: torch brings its own HIP runtime; loaded after the library's, it finds no device)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND = {"bls12_377_g1": "g1_377", "bls12_377_g2": "g2_377", "bw6_761_g1": "761", "bw6_761_g2": "761"}
AFF = {"bls12_377_g1": 12, "bls12_377_g2": 24, "bw6_761_g1": 24, "bw6_761_g2": 24}


def _threads():
    return max(2, min(32, os.cpu_count() or 2))


# ------------------------------------------------------------------------------------------------ CPU: the join
def test_join_windows_matches_big_integer_sum():
    """msm_*_join_windows is host arithmetic (no device): total = sum_g 2^bit_lo[g] P_g for affine records, identities included."""
    from celo_bls_snark_rs_amd import ffi, synthetic as syn
    rng = ecc.SplitMix64(4242)
    for group, cur, gen, pack, n64, p in (("bls12_377_g1", ecc.E1_377, ecc.G1_377, co.pack_g1_377, 6, ecc.Q377),
                                          ("bls12_377_g2", ecc.E2_377, ecc.G2_377, co.pack_g2_377, 12, ecc.Q377),
                                          ("bw6_761_g1", ecc.E1_761, syn.BW6_G1_POINT, co.pack_761, 12, ecc.Q761)):
        one = co.to_mont([1], p)[0] if group != "bls12_377_g2" else np.concatenate([co.to_mont([1], p)[0], np.zeros(6, dtype=np.uint64)])
        for bits in ([0], [0, 32, 64, 96, 128, 160, 192, 222], [0, 0, 45], [0, 16, 31]):
            pts = [cur.mul(gen, rng.next() | 1) for _ in bits]
            if len(pts) > 2:
                pts[1] = None                                              # an empty shard: the identity record (ZZ = 0)
            recs = np.zeros((len(bits), 4 * n64), dtype=np.uint64)
            for g, P in enumerate(pts):
                if P is None:
                    continue
                xy, _ = pack([P])
                recs[g, :2 * n64] = xy.reshape(-1)
                recs[g, 2 * n64:3 * n64] = one
                recs[g, 3 * n64:] = one
            want = None
            for g, P in enumerate(pts):
                if P is not None:
                    want = cur.add(want, cur.mul(P, 1 << bits[g]))
            got = co.jac_to_affine(ffi.join_windows(group, recs, bits), KIND[group])
            assert got == want, (group, bits)
    with pytest.raises(RuntimeError):                                      # descending bit offsets are refused
        ffi.join_windows("bls12_377_g1", np.zeros((2, 24), dtype=np.uint64), [16, 0])


# ------------------------------------------------------------------------------------------------ 1 GPU: the window partition
@pytest.mark.gpu
@pytest.mark.parametrize("log_n,shards", [(8, [1, 2, 5]), (13, [3, 23, 30]), (15, [2, 8]), (17, [8]), (19, [3, 8, 16]), (20, [8])])
def test_window_partition_g1_vs_oracle(gpu, log_n, shards):
    """Every window layout of the plain path: c = 11 (23 windows), c = 15 (17), c = 16 with mixed widths (14 x 16 + 2 x 15 bits), with
    shard counts that divide the windows, that do not, and that exceed them (surplus shards are empty)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    import torch
    n = 1 << log_n
    pts = syn.device_points("bls12_377_g1", n, 7000 + log_n)
    sc = syn.uniform_scalars("bls12_377_g1", n, 7100 + log_n)
    sc[:6] = H.scalars_np([0, 1, ecc.R377 - 1, 1 << 64, (1 << 253) - 1 - (1 << 200), 2], 4)        # edge scalars incl. every top-window bit
    h = pts.cpu().numpy().view(np.uint64).reshape(n, 12)
    exp = co.jac_to_affine(co.msm("bls12_377_g1", h, None, sc, threads=_threads()), "g1_377")
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    for k in shards:
        got = gpu.msm_multi_windows_dev("bls12_377_g1", [0] * k, [pts.data_ptr()] * k, None, [d_sc.data_ptr()] * k, n)
        assert co.jac_to_affine(got, "g1_377") == exp, (log_n, k)
    if log_n <= 15:                                                        # the host-pointer form stages the input once per shard
        inf = np.zeros(n, dtype=np.uint8); inf[7] = 1
        exp_i = co.jac_to_affine(co.msm("bls12_377_g1", h, inf, sc, threads=_threads()), "g1_377")
        assert co.jac_to_affine(gpu.msm_multi_windows("bls12_377_g1", [0, 0, 0], h, inf, sc), "g1_377") == exp_i


@pytest.mark.gpu
def test_window_partition_subgroup_entries_and_other_groups(gpu):
    """The GLV split's 8 windows of 16 bits over 2 n points (G1 and G2 subgroup entries), G2 and BW6-761 plain."""
    from celo_bls_snark_rs_amd import synthetic as syn
    import torch
    for group, n, sub, ks in (("bls12_377_g1", 1 << 15, True, [3, 8]), ("bls12_377_g2", 1 << 14, True, [4]), ("bls12_377_g2", 5000, False, [5]),
                              ("bw6_761_g1", 6000, False, [4, 29]), ("bw6_761_g2", 1 << 19, False, [8])):
        pts = syn.device_points(group, n, 7300 + n)
        sc = syn.uniform_scalars(group, n, 7301 + n)
        h = pts.cpu().numpy().view(np.uint64).reshape(n, AFF[group])
        exp = co.jac_to_affine(co.msm(group, h, None, sc, threads=_threads()), KIND[group])
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        for k in ks:
            got = gpu.msm_multi_windows_dev(group, [0] * k, [pts.data_ptr()] * k, None, [d_sc.data_ptr()] * k, n, subgroup=sub)
            assert co.jac_to_affine(got, KIND[group]) == exp, (group, n, k)


@pytest.mark.gpu
def test_window_shard_records_join_like_the_ranks_do(gpu):
    """One process per GPU: every rank computes msm_*_window_shard_dev(shard = rank), the records are gathered, every rank joins.
    Here the 'ranks' run one after the other on device 0."""
    from celo_bls_snark_rs_amd import synthetic as syn
    import torch
    for group, n, sub in (("bls12_377_g1", 1 << 16, False), ("bls12_377_g1", 1 << 16, True), ("bw6_761_g1", 4000, False), ("bls12_377_g2", 3000, False)):
        pts = syn.device_points(group, n, 7500)
        sc = syn.uniform_scalars(group, n, 7501)
        h = pts.cpu().numpy().view(np.uint64).reshape(n, AFF[group])
        exp = co.jac_to_affine(co.msm(group, h, None, sc, threads=_threads()), KIND[group])
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        for world in (1, 2, 4, 8):
            recs, bits = [], []
            for r in range(world):
                rec, b = gpu.msm_window_shard_dev(group, pts.data_ptr(), 0, d_sc.data_ptr(), n, r, world, subgroup=sub)
                recs.append(rec); bits.append(b)
            assert bits == sorted(bits) and bits[0] == 0
            assert co.jac_to_affine(gpu.join_windows(group, np.stack(recs), bits), KIND[group]) == exp, (group, world)


# ------------------------------------------------------------------------------------------------ >= 2 devices
def _need_two(gpu):
    n = gpu.device_count()
    if n < 2:
        pytest.skip("needs >= 2 MI355X devices (this box has %d)" % n)
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("group,log_n", [("bls12_377_g1", 17), ("bls12_377_g1", 20), ("bls12_377_g2", 17), ("bw6_761_g1", 17)])
def test_index_range_shards_on_distinct_devices(gpu, group, log_n):
    """msm_*_multi_dev with shard d RESIDENT on device d (not device 0 listed twice): 2^log_n terms per device vs the oracle."""
    import torch
    from celo_bls_snark_rs_amd import synthetic as syn
    ndev = min(_need_two(gpu), 8)
    n = 1 << log_n
    bases, scalars, hb, hs = [], [], [], []
    for d in range(ndev):
        with torch.cuda.device(d):
            gpu.use_device(d)
            pts = syn.device_points(group, n, 8000 + 13 * d + log_n)
            sc = syn.uniform_scalars(group, n, 8001 + 13 * d + log_n)
            bases.append(pts); scalars.append(torch.from_numpy(sc.view(np.int64)).cuda(d))
            hb.append(pts.cpu().numpy().view(np.uint64).reshape(n, AFF[group])); hs.append(sc)
    gpu.use_device(0)
    got = gpu.msm_multi_dev(group, list(range(ndev)), [b.data_ptr() for b in bases], None, [s.data_ptr() for s in scalars], [n] * ndev)
    exp = co.msm(group, np.concatenate(hb), None, np.concatenate(hs), threads=_threads())
    assert co.jac_to_affine(got, KIND[group]) == co.jac_to_affine(exp, KIND[group])
    # host-pointer form: the library cuts and stages the index ranges itself
    got_h = gpu.msm_multi(group, list(range(ndev)), np.concatenate(hb), None, np.concatenate(hs))
    assert co.jac_to_affine(got_h, KIND[group]) == co.jac_to_affine(exp, KIND[group])


@pytest.mark.gpu
@pytest.mark.parametrize("group,log_n,sub", [("bls12_377_g1", 20, False), ("bls12_377_g1", 20, True), ("bls12_377_g2", 17, False), ("bw6_761_g1", 17, False)])
def test_window_partition_on_distinct_devices(gpu, group, log_n, sub):
    """msm_*_multi_windows_dev with one replica of the n terms on every device."""
    import torch
    from celo_bls_snark_rs_amd import synthetic as syn
    ndev = min(_need_two(gpu), 8)
    n = 1 << log_n
    gpu.use_device(0)
    with torch.cuda.device(0):
        pts0 = syn.device_points(group, n, 8100 + log_n)
    sc = syn.uniform_scalars(group, n, 8101 + log_n)
    h = pts0.cpu().numpy().view(np.uint64).reshape(n, AFF[group])
    reps_b = [pts0] + [torch.from_numpy(h.view(np.int64)).cuda(d) for d in range(1, ndev)]
    reps_s = [torch.from_numpy(sc.view(np.int64)).cuda(d) for d in range(ndev)]
    got = gpu.msm_multi_windows_dev(group, list(range(ndev)), [b.data_ptr() for b in reps_b], None, [s.data_ptr() for s in reps_s], n, subgroup=sub)
    exp = co.msm(group, h, None, sc, threads=_threads())
    assert co.jac_to_affine(got, KIND[group]) == co.jac_to_affine(exp, KIND[group])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--scaling", "weak"], ["--scaling", "strong"], ["--scaling", "strong", "--partition", "index"],
                                   ["--config", "3", "--batches", "512"], ["--in-process"]])
def test_bench_two_ranks_over_rccl(gpu, extra):
    """`bench.py --gpus 2` launches its own two ranks over RCCL (backend nccl), one per device; the line must say so and must have
    passed the full-size parity check of the folded / joined result against the CPU port."""
    _need_two(gpu)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CELO_BENCH_BACKEND", None); env.pop("CELO_BENCH_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    if "--in-process" in extra:
        assert line["launch"]["mode"].startswith("in-process") and sorted(set(line["launch"]["devices"])) == [0, 1]
    else:
        assert line["launch"]["rccl_world"] == 2 and "nccl" in line["launch"]["backend"]
    assert line.get("parity", {}).get("checked") is True
