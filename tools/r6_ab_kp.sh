#!/bin/bash
# same-box A/B, config 4: the BW6-761 units with the K p tables as immediates (v1 = the shipped library) against the pointer form of rounds 1-5
# (ptr: unit_761 + unit_761_aux built with -DCELO_KP_PTR_TABLES, both with -amdgpu-long-branch-factor=0)
export AB_CMD='python bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],3), d[\"roofline\"][\"note\"][-100:])"'
export AB_TAIL=1
bash tools/ab_generic.sh v1 ptr   # (and v1 op: the opaque-pointer form, built the same way with -DCELO_KP_OPAQUE_TABLES before it became the default)
