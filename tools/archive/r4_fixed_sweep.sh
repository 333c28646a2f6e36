#!/bin/bash
# fixed-base MSM: window-size sweep on config 4 (BW6-761 G1, 2^21) and config 2 (BLS12-377 G1, 2^20), beside the variable-base entry
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); fb=l['config'].get('fixed_base')
print('$1', round(l['ms_per_step'],2), 'ms', '%.3g' % l['value'], l['unit'], (fb or ''), l['roofline']['note'].split('MSM stream: ')[1])"; }
for cfg in 4 2; do
  python bench.py --config $cfg --no-cpu-baseline --no-pairing --steps 5 2>/dev/null | tail -1 | line "cfg$cfg variable"
  for c in ${CS:-16 17 18 19 20 21 22}; do
    python bench.py --config $cfg --fixed-base --fixed-window-bits $c --no-cpu-baseline --steps 5 2>/dev/null | tail -1 | line "cfg$cfg fixed c=$c"
  done
done
