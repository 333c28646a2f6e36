#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_2p17 -o t --output-format csv -- python $ROOT/bench.py --log-n 17 --steps 10 --warmup 3 --no-cpu-baseline --no-pairing > $ROOT/gpurun_out/prof_2p17.log 2>&1
cd $ROOT && python - <<'PY'
import csv, json
rows=list(csv.DictReader(open('gpurun_out/prof_2p17/t_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name'].replace('void celo::','').split('(')[0][:50]
    if 'G1_377' in n or 'k_digits' in n or 'rocclr' in n:
        print("%-52s %4s %10.1f us"%(n,r['Calls'],float(r['AverageNs'])/1e3))
print(open('gpurun_out/prof_2p17.log').read()[-1500:])
PY
