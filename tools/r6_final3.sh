#!/bin/bash
# after the opaque-pointer K p tables of the 28-limb field: the whole GPU suite, the reproducer of the long-branch hang, config 4's profile passes,
# the groups table, then every configuration's bench line
O=gpurun_out/r6_final3; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
bash tools/repro_combine/build.sh > $O/repro_build.txt 2>&1; timeout 300 celo-bls-snark-rs_amd/build/repro_combine > $O/repro_combine.txt 2>&1; echo "repro rc=$?"; tail -2 $O/repro_combine.txt
bash tools/profile_bench.sh r6_cfg4 "python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline" > $O/profile_cfg4.log 2>&1
bash tools/profile_bench.sh r6_groups "python $GRAFT_REPO_ROOT/tools/bench_groups.py 20" > $O/profile_groups.log 2>&1
find gpurun_out -name "*_kernel_trace.csv" -size +8M -delete
timeout 1500 python bench.py --all-configs > $O/bench_all.jsonl 2> $O/bench_all.err; echo "bench rc=$?"; cut -c1-150 $O/bench_all.jsonl
timeout 300 python tools/bench_latency.py 2>/dev/null | tail -1 | cut -c1-300
