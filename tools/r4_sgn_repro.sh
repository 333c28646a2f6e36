#!/bin/bash
# Round-3 open finding: (1) the in-kernel comparison of the two forms of the Fq2 pass on identical operands (tools/repro_mul4k.hip);
# (2) the library built with -DCELO_MUL4K_SGN=true on the inputs that failed in round 3.
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== repro_mul4k 2^17 lanes x 64"; timeout 300 celo-bls-snark-rs_amd/build/repro_mul4k 17 64 | head -40
echo "== repro_mul4k 2^18 lanes x 256"; timeout 600 celo-bls-snark-rs_amd/build/repro_mul4k 18 256 | head -40
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L /tmp/lib_main.so
cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_sgn.so $L
echo "== signed library: G2 tests"
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -q -k "g2" 2>&1 | tail -8
cp /tmp/lib_main.so $L
