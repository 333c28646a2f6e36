#!/usr/bin/env python3
"""Copies one sweep's results (gpurun_out/r3/, written by tools/archive/r3_sweep.sh on the GPU box) into profiles/r3_* and runs
tools/summarise_profile.py over its rocprofv3 output sets.  Run on the host after the gpurun call returns."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3")
DST = os.path.join(ROOT, "profiles")
for t in ("r3", "r3_groups", "r3_cfg3", "r3_cfg4", "r3_cfg5", "r3_pairing"):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarise_profile.py"), t])
for f in sorted(os.listdir(SRC)):
    if f.endswith(".json"):
        body = open(os.path.join(SRC, f)).read().strip()
        if not body:
            print("EMPTY", f)
            continue
        json.loads(body.splitlines()[-1])
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, "r3_" + f))
for g in ("g1", "g2"):
    p = os.path.join(SRC, "glv_%s.jsonl" % g)
    rows = {}
    for line in open(p):
        line = line.strip()
        if not line:
            continue
        logn, body = line.split(" ", 1)
        rows["2^" + logn] = json.loads(body)
    json.dump({"tool": "tools/bench_glv.py %s 14 16 17 18 20" % g, "rows": rows}, open(os.path.join(DST, "r3_glv_%s.json" % g), "w"), indent=1)
pm = os.path.join(ROOT, "gpurun_out", "r3_pair_pmc", "summary.txt")
if os.path.exists(pm):
    shutil.copy(pm, os.path.join(DST, "r3_pairing_issue_counters.txt"))
print("ok")
