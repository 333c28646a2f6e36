#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, every bench configuration, the N = 2 launcher / in-process forms on one shared GPU
# (both partitions), the fixed-base and window-shard tables, the per-group tables, the latency / Seam A / decode / hash / NTT / criterion
# tools and the rocprofv3 profile sets of round 4 (kernel trace + separate FETCH / WRITE / SQ / clock passes).
# Everything lands under gpurun_out/r4/ and gpurun_out/prof_r4*/ ; tools/collect_r4.py copies the summaries into profiles/ on the host.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py --config 2 --steps 20 --warmup 3 2>$OUT/cfg2.err | tail -1 > $OUT/bench_cfg2.json
python bench.py --config 3 --steps 10 --warmup 2 2>$OUT/cfg3.err | tail -1 > $OUT/bench_cfg3.json
python bench.py --config 4 --steps 10 --warmup 2 2>$OUT/cfg4.err | tail -1 > $OUT/bench_cfg4.json
python bench.py --config 4 --steps 10 --warmup 2 --fixed-base 2>$OUT/cfg4f.err | tail -1 > $OUT/bench_cfg4_fixed_base.json
python bench.py --config 4 --steps 10 --warmup 2 --witness-like --no-cpu-baseline 2>$OUT/cfg4w.err | tail -1 > $OUT/bench_cfg4_witness_like.json
python bench.py --config 4 --steps 10 --warmup 2 --witness-like --fixed-base --no-cpu-baseline 2>$OUT/cfg4wf.err | tail -1 > $OUT/bench_cfg4_witness_like_fixed_base.json
python bench.py --config 5 --steps 10 --warmup 2 2>$OUT/cfg5.err | tail -1 > $OUT/bench_cfg5.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --scaling strong 2>$OUT/n2s.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_strong_windows_shared_gpu.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --scaling strong --partition index 2>$OUT/n2si.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_strong_index_shared_gpu.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/n2w.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_weak_shared_gpu.json
python bench.py --gpus 2 --in-process --devices 0,0 --steps 10 --warmup 2 --scaling strong 2>$OUT/inproc.err | tail -1 > $OUT/bench_cfg2_gpus2_in_process_strong_windows_shared_gpu.json
python bench.py --log-n 17 --steps 20 --warmup 3 --no-pairing 2>$OUT/2p17.err | tail -1 > $OUT/bench_cfg2_2p17_index_range_shard.json
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tools/bench_windows.py g1 20 --world 2,4,8 2>/dev/null | tail -1 > $OUT/windows_g1_2p20.json
python tools/bench_windows.py g1 20 --world 8 --subgroup 2>/dev/null | tail -1 > $OUT/windows_g1_2p20_subgroup.json
python tools/bench_windows.py g2 20 --world 8 2>/dev/null | tail -1 > $OUT/windows_g2_2p20.json
python tools/bench_windows.py bw6 20 --world 8 2>/dev/null | tail -1 > $OUT/windows_bw6_2p20.json
python tools/bench_hybrid.py 20 2>/dev/null > $OUT/hybrid_g1_2p20.jsonl
bash tools/archive/r4_fixed_sweep.sh > $OUT/fixed_sweep.txt 2>&1
python tools/ab_cfg3_chains.py 2>/dev/null | tail -1 > $OUT/cfg3_chains.json
python tools/bench_criterion_shapes.py 2>/dev/null | tail -1 > $OUT/criterion_shapes.json
python tools/bench_glv.py g1 14 16 17 18 20 2>/dev/null > $OUT/glv_g1.jsonl
python tools/bench_glv.py g2 14 16 17 18 20 2>/dev/null > $OUT/glv_g2.jsonl
for l in 14 17 18 20; do python tools/bench_groups.py $l 2>/dev/null | tail -1 > $OUT/groups_2p$l.json; done
python tools/bench_latency.py > $OUT/latency.json 2>$OUT/latency.err
python tools/bench_seam_a_strict.py 4096 > $OUT/seam_a_strict.json 2>$OUT/seam_a_strict.err
python tools/bench_pairing.py 2048 4096 10240 20480 40960 81920 2>/dev/null | tail -1 > $OUT/pairing_sizes.json
python tools/bench_decompress.py 2>/dev/null | tail -1 > $OUT/decompress.json
python tools/bench_hash.py 2>/dev/null | tail -1 > $OUT/hash.json
python tools/bench_ntt.py 2>/dev/null | tail -1 > $OUT/ntt.json
celo-bls-snark-rs_amd/build/repro_mul4k 17 64 > $OUT/repro_mul4k.txt 2>&1
celo-bls-snark-rs_amd/build/repro_acc 16 24 2 2048 > $OUT/repro_acc.txt 2>&1
celo-bls-snark-rs_amd/build/repro_acc_uni 16 24 2 2048 > $OUT/repro_acc_uniform.txt 2>&1
bash tools/archive/r4_profiles.sh r4 > $OUT/profiles.log 2>&1
bash tools/archive/r4_pair_pmc.sh > $OUT/pair_pmc.log 2>&1
for f in $OUT/*.json; do echo "== $f"; head -c 300 $f; echo; done
