#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_pairing_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_pairing.py 81920 2>/dev/null | tail -1
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.')
from celo_bls_snark_rs_amd import ffi
import bench
ffi.init(0)
r = bench.pairing_leg(ffi, check_oracle=False)
print({k: r[k] for k in ("value", "miller_ms", "final_exp_ms", "device_ms")})
PY
