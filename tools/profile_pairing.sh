#!/bin/bash
# rocprofv3 kernel trace + FETCH/WRITE PMC passes of the pairing throughput tool (tools/bench_pairing.py) -> gpurun_out/prof_<tag>/
TAG=${1:-r2_pairing}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/profile_bench.sh $TAG "python $ROOT/tools/bench_pairing.py 81920"
