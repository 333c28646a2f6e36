#!/bin/bash
# Runs on the GPU box (via gpurun): every bench configuration, the latency / Seam A / pairing-size tools and the two rocprofv3
# profile sets, all under gpurun_out/sweep/ and gpurun_out/prof_<tag>/ ; copied into profiles/ afterwards (see README "Measuring").
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sweep
mkdir -p $OUT
cd $ROOT
python bench.py --config 2 --steps 20 --warmup 3 2>$OUT/cfg2.err | tail -1 > $OUT/bench_cfg2.json
python bench.py --config 3 --steps 10 --warmup 2 2>$OUT/cfg3.err | tail -1 > $OUT/bench_cfg3.json
python bench.py --config 4 --steps 10 --warmup 2 2>$OUT/cfg4.err | tail -1 > $OUT/bench_cfg4.json
python bench.py --config 4 --steps 10 --warmup 2 --witness-like --no-cpu-baseline 2>$OUT/cfg4w.err | tail -1 > $OUT/bench_cfg4_witness_like.json
python bench.py --config 5 --steps 10 --warmup 2 2>$OUT/cfg5.err | tail -1 > $OUT/bench_cfg5.json
python tools/bench_latency.py > $OUT/latency.json 2>$OUT/latency.err
python tools/bench_seam_a_strict.py 4096 > $OUT/seam_a_strict.json 2>$OUT/seam_a_strict.err
python tools/bench_pairing.py 2048 4096 10240 20480 40960 81920 2>/dev/null | tail -1 > $OUT/pairing_sizes.json
bash tools/profile_bench.sh r2 > $OUT/profile_r2.log 2>&1
bash tools/profile_pairing.sh r2_pairing > $OUT/profile_r2_pairing.log 2>&1
for f in $OUT/*.json; do echo "== $f"; head -c 400 $f; echo; done
