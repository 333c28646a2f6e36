#!/bin/bash
# round 6, second GPU call: the new distinct-data pairing tests, Seam A, then the bench lines of every configuration in one run
O=gpurun_out/r6_second; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pairing_gpu.py tests/test_seam_a.py -m gpu -x -q > $O/pytest_pairing.txt 2>&1; echo "pytest pairing rc=$?"; tail -5 $O/pytest_pairing.txt
timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "cfg5 or cfg3" > $O/pytest_cfg.txt 2>&1; echo "pytest cfg rc=$?"; tail -3 $O/pytest_cfg.txt
timeout 1500 python bench.py --all-configs > $O/bench_all.jsonl 2> $O/bench_all.err; echo "bench rc=$?"; cut -c1-400 $O/bench_all.jsonl; tail -5 $O/bench_all.err
