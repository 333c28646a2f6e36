#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, every bench configuration, the N = 2 launcher / in-process forms on one shared GPU,
# the per-group tables, the latency / Seam A / decode / hash / NTT tools and the rocprofv3 profile sets of round 3.
# Everything lands under gpurun_out/r3/ and gpurun_out/prof_r3*/ ; tools/summarise_profile.py and a copy into profiles/ follow on the host.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py --config 2 --steps 20 --warmup 3 2>$OUT/cfg2.err | tail -1 > $OUT/bench_cfg2.json
python bench.py --config 3 --steps 10 --warmup 2 2>$OUT/cfg3.err | tail -1 > $OUT/bench_cfg3.json
python bench.py --config 4 --steps 10 --warmup 2 2>$OUT/cfg4.err | tail -1 > $OUT/bench_cfg4.json
python bench.py --config 4 --steps 10 --warmup 2 --witness-like --no-cpu-baseline 2>$OUT/cfg4w.err | tail -1 > $OUT/bench_cfg4_witness_like.json
python bench.py --config 5 --steps 10 --warmup 2 2>$OUT/cfg5.err | tail -1 > $OUT/bench_cfg5.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --scaling strong 2>$OUT/n2s.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_strong_shared_gpu.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/n2w.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_weak_shared_gpu.json
python bench.py --gpus 2 --in-process --devices 0,0 --steps 10 --warmup 2 --scaling strong 2>$OUT/inproc.err | tail -1 > $OUT/bench_cfg2_gpus2_in_process_strong_shared_gpu.json
python bench.py --log-n 17 --steps 20 --warmup 3 --no-pairing 2>$OUT/2p17.err | tail -1 > $OUT/bench_cfg2_2p17_strong_scaling_shard.json
python bench.py --log-n 17 --steps 20 --warmup 3 --no-pairing --no-cpu-baseline --subgroup-points 2>/dev/null | tail -1 > $OUT/bench_cfg2_2p17_strong_scaling_shard_subgroup_entry.json
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tools/bench_glv.py g1 14 16 17 18 20 2>/dev/null > $OUT/glv_g1.jsonl
python tools/bench_glv.py g2 14 16 17 18 20 2>/dev/null > $OUT/glv_g2.jsonl
for l in 14 17 18 20; do python tools/bench_groups.py $l 2>/dev/null | tail -1 > $OUT/groups_2p$l.json; done
python tools/bench_latency.py > $OUT/latency.json 2>$OUT/latency.err
python tools/bench_seam_a_strict.py 4096 > $OUT/seam_a_strict.json 2>$OUT/seam_a_strict.err
python tools/bench_pairing.py 2048 4096 10240 20480 40960 81920 2>/dev/null | tail -1 > $OUT/pairing_sizes.json
python tools/bench_decompress.py 2>/dev/null | tail -1 > $OUT/decompress.json
python tools/bench_hash.py 2>/dev/null | tail -1 > $OUT/hash.json
python tools/bench_ntt.py 2>/dev/null | tail -1 > $OUT/ntt.json
bash tools/archive/r3_profiles.sh r3 > $OUT/profiles.log 2>&1
bash tools/archive/r3_pair_pmc.sh > $OUT/pair_pmc.log 2>&1
for f in $OUT/*.json; do echo "== $f"; head -c 300 $f; echo; done
