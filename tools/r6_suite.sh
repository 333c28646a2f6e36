#!/bin/bash
# the whole -m gpu suite under its per-test wall-clock bounds; output merged back under gpurun_out/r6_suite
O=gpurun_out/r6_suite; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
