#!/bin/bash
# Runs on the GPU box: one rocprofv3 kernel-trace pass of a command (default: the bench), prints the top of the per-kernel table.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CMD=${1:-"python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/qs && rocprofv3 --kernel-trace --stats -d /tmp/qs -o t --output-format csv -- $CMD > /tmp/qs.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/qs/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${2:-28}]: print("%-60s %6s %12.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
