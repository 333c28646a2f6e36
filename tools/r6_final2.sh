#!/bin/bash
# round 6, final measurement set, part 2: every configuration's bench line with its CPU baseline (traffic from profiles/r6_*_traffic.json)
O=gpurun_out/r6_final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py --all-configs > $O/bench_all.jsonl 2> $O/bench_all.err; echo "bench rc=$?"; cut -c1-160 $O/bench_all.jsonl
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
