#!/bin/bash
# batched-affine pre-levels (BW6-761): parity tests, then config 4's bench line over the variants (same box), then a kernel trace
O=gpurun_out/r6_ba; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_msm_ba_gpu.py -m gpu -x -q > $O/pytest_ba.txt 2>&1; echo "pytest ba rc=$?"; tail -3 $O/pytest_ba.txt
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$tag.json 2> $O/bench_$tag.err; rc=$?
  python - <<P
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", "rc=$rc", round(d["ms_per_step"],3), d["roofline"]["note"][-100:])
except Exception as e:
    print("$tag", "rc=$rc", "no line", e)
P
}
run ba0 CELO_BA=0
run occ1_k3 CELO_BA_OCC=1
run occ2_k3 CELO_BA_OCC=2
run occ1_k2 CELO_BA_OCC=1 CELO_BA_LEVELS=2
run occ2_k2 CELO_BA_OCC=2 CELO_BA_LEVELS=2
run occ1_k3_r2 CELO_BA_OCC=1 CELO_BA_ROUNDS=2
run occ2_k3_r2 CELO_BA_OCC=2 CELO_BA_ROUNDS=2
run occ1_k4 CELO_BA_OCC=1 CELO_BA_LEVELS=4
run ba0_again CELO_BA=0
(cd /tmp && CELO_BA_OCC=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
find $O/trace -name "*kernel_trace.csv" -delete
