#!/bin/bash
O=gpurun_out/r6_ba_prof; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
find $O/trace -name "*kernel_stats*" | head; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
