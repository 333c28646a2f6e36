#!/bin/bash
# World-size-8 dry runs on a ONE-GPU box: eight ranks (gloo, host-staged exchange) sharing device 0, every partition and
# config that the driver's 8-GPU SCALE run or a caller could take, each with its parity check; lines -> gpurun_out/r5_w8/*.json
#   tools/world8.sh [steps]   (TAG=r6 by default: lines -> gpurun_out/<TAG>_w8/*.json)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/${TAG:-r6}_w8; mkdir -p $OUT
STEPS=${1:-3}
export CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 OMP_NUM_THREADS=8
run() { name=$1; shift; echo "== $name: $*"; timeout 900 python bench.py --gpus 8 --steps $STEPS --warmup 1 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "rc=$? $(tail -c 300 $OUT/$name.json | head -c 300)"; }
run weak_cfg2 --scaling weak
run strong_windows_cfg2 --scaling strong
run strong_index_cfg2 --scaling strong --partition index
run cfg3 --config 3 --scaling strong
run cfg4_weak --config 4 --log-n 18
run cfg4_strong_2p21 --config 4 --scaling strong --log-n 21
run cfg5 --config 5 --scaling strong
unset CELO_BENCH_BACKEND CELO_BENCH_DEVICE
echo "== in_process"; timeout 900 python bench.py --gpus 8 --steps $STEPS --warmup 1 --in-process --devices 0,0,0,0,0,0,0,0 > $OUT/in_process_cfg2.json 2> $OUT/in_process_cfg2.err; echo "rc=$? $(tail -c 300 $OUT/in_process_cfg2.json)"
echo "== in_process strong windows"; timeout 900 python bench.py --gpus 8 --steps $STEPS --warmup 1 --in-process --devices 0,0,0,0,0,0,0,0 --scaling strong > $OUT/in_process_strong_cfg2.json 2> $OUT/in_process_strong_cfg2.err; echo "rc=$? $(tail -c 300 $OUT/in_process_strong_cfg2.json)"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "n_gpus", d["n_gpus"], "value %.4g" % d["value"], d["unit"], "ms %.3f" % d["ms_per_step"], "parity", d.get("parity", {}).get("checked"), "strong" in d and ("strong %.4g" % d["strong"]["value"]) or "")
    except Exception as e:
        print(f, "NO LINE", e)
PY
