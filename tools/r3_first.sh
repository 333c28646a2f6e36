#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, the default bench line, the self-launched 2-rank run on one shared GPU,
# the in-process two-engine run, and the per-group timing table at the strong-scaling shard sizes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 3 2>$OUT/cfg2.err | tail -1 > $OUT/bench_cfg2.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 --scaling strong 2>$OUT/n2_strong.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_strong_shared_gpu.json
CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/n2_weak.err | tail -1 > $OUT/bench_cfg2_gpus2_self_launched_weak_shared_gpu.json
python bench.py --gpus 2 --in-process --devices 0,0 --steps 10 --warmup 2 --scaling strong 2>$OUT/inproc.err | tail -1 > $OUT/bench_cfg2_gpus2_in_process_strong_shared_gpu.json
for l in 14 16 17 18 20; do python tools/bench_groups.py $l 2>/dev/null | tail -1 > $OUT/groups_2p$l.json; done
for f in $OUT/*.json; do echo "== $f"; head -c 600 $f; echo; done
tail -3 $OUT/*.err
