#!/bin/bash
# same-box A/B: the two G1 units with and without the SLP vectorizer (build_ab/libcelo_bls_amd_slp.so against build/), config 2, alternating
export AB_CMD='python bench.py --no-cpu-baseline --no-pairing --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],4), d[\"roofline\"][\"note\"][-95:])"'
export AB_TAIL=1
bash tools/ab_generic.sh v1 slp
bash tools/ab_generic.sh v1 slp
