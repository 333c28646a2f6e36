#!/bin/bash
# Same-box A/B of library variants on any command: AB_CMD='python tools/bench_groups.py 20' ab_generic.sh v1 v3
# (v1 = the library in build/; others = build_ab/libcelo_bls_amd_<v>.so).  Two rounds, alternating.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_v1.so
for round in 1 2; do
  for v in "$@"; do
    cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_$v.so $L
    echo "== $v round $round"
    bash -c "$AB_CMD" 2>/dev/null | tail -${AB_TAIL:-1}
  done
done
cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_v1.so $L
