#!/bin/bash
# Same-box A/B of library variants (gpurun boxes differ by a few per cent): every build_ab/libcelo_bls_amd_<v>.so is put in place of the
# library in turn, twice round, and the verify-shaped pairing leg of bench.py is timed.  Usage (on the GPU box): ab_pairing.sh v0 v1 v2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_v1.so
for round in 1 2; do
  for v in "$@"; do
    cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_$v.so $L
    python - <<PY
import sys
sys.path.insert(0, '.')
from celo_bls_snark_rs_amd import ffi
import bench
ffi.init(0)
r = bench.pairing_leg(ffi, check_oracle=False)
r = bench.pairing_leg(ffi, check_oracle=False)
print("$v round $round", {k: round(r[k], 3) for k in ("miller_ms", "final_exp_ms", "device_ms")}, "%.3e" % r["value"])
PY
  done
done
for v in "$@"; do
  cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_$v.so $L
  echo "== tests on $v"; timeout 900 python -m pytest tests/test_pairing_gpu.py -m gpu -x -q 2>&1 | tail -1
done
cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_v1.so $L
