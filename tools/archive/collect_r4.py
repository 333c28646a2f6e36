#!/usr/bin/env python3
"""Copies one sweep's results (gpurun_out/r4/, written by tools/archive/r4_sweep.sh on the GPU box) into profiles/r4_* and runs
tools/summarise_profile.py over its rocprofv3 output sets.  Run on the host after the gpurun call returns."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r4")
DST = os.path.join(ROOT, "profiles")
for t in ("r4", "r4_groups", "r4_cfg3", "r4_cfg4", "r4_cfg5", "r4_pairing"):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarise_profile.py"), t])
for f in sorted(os.listdir(SRC)):
    if f.endswith(".json"):
        body = open(os.path.join(SRC, f)).read().strip()
        if not body:
            print("EMPTY", f)
            continue
        json.loads(body.splitlines()[-1])
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, "r4_" + f))
for g in ("g1", "g2"):
    p = os.path.join(SRC, "glv_%s.jsonl" % g)
    rows = {}
    for line in open(p):
        line = line.strip()
        if not line:
            continue
        logn, body = line.split(" ", 1)
        rows["2^" + logn] = json.loads(body)
    json.dump({"tool": "tools/bench_glv.py %s 14 16 17 18 20" % g, "rows": rows}, open(os.path.join(DST, "r4_glv_%s.json" % g), "w"), indent=1)
for f in ("fixed_sweep.txt", "repro_mul4k.txt", "repro_acc.txt", "repro_acc_uniform.txt", "hybrid_g1_2p20.jsonl"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, "r4_" + f))
for f, name in (("sgn_sites3.txt", "r4_signed_pass_sites.txt"), ("sgn_hypothesis.txt", "r4_signed_pass_zero_test_variants.txt"), ("repro_acc_flags.txt", "r4_repro_acc_compiler_flags.txt"),
                ("ab_occ1.txt", "r4_ab_decode_hash_one_wave.txt"), ("windows_sweep.txt", "r4_windows_piece_length_sweep.txt")):
    for d in (os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "gpurun_out", "r4")):
        if os.path.exists(os.path.join(d, f)):
            shutil.copy(os.path.join(d, f), os.path.join(DST, name))
pm = os.path.join(ROOT, "gpurun_out", "r4_pair_pmc", "summary.txt")
if os.path.exists(pm):
    shutil.copy(pm, os.path.join(DST, "r4_pairing_issue_counters.txt"))
print("ok")
