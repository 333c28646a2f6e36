#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/r3e
timeout 1200 python -m pytest tests/test_batch_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "batch or cfg3 or cfg1" 2>&1 | tail -8
python bench.py --config 3 --steps 10 --warmup 2 2>gpurun_out/r3e/cfg3.err | tail -1 > gpurun_out/r3e/bench_cfg3_gls.json
CELO_NO_GLS=1 python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3e/bench_cfg3_nogls.json
for f in gpurun_out/r3e/*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['roofline']['note'])"; done
tail -3 gpurun_out/r3e/cfg3.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_gls -o t --output-format csv -- python $ROOT/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $ROOT && python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_gls/t_kernel_stats.csv')))
for r in rows[:14]:
    n=r['Name'].replace('void celo::','').split('(')[0][:50]
    print("%-52s %4s %10.1f us"%(n,r['Calls'],float(r['AverageNs'])/1e3))
PY
