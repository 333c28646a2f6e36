#!/bin/bash
# window-shard tuning sweep (same box): piece length rule, lane-bitsum threshold, side-stream conversion
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
run() { echo "== $1"; env $1 python tools/bench_windows.py ${GRP:-g1} ${LOGN:-20} --world ${WORLD:-8} $SUB 2>/dev/null | python -c '
import json,sys
r=json.loads(sys.stdin.readline()); w=[v for k,v in r.items() if k.startswith("world_")][0]
print("single", r["single_call_ms"], "worst", w["worst_shard_call_ms"], "join", w["join_ms"], "bound", w["strong_scaling_bound"])
for i in (0, len(w["shards"])-1):
  s=w["shards"][i]; print("  shard", i, {k:s[k] for k in ("call_ms","dev_ms","conv","sort","acc","red","windows")})'; }
for v in "$@"; do run "$v"; done
