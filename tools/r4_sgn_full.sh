#!/bin/bash
# the whole -m gpu suite against the library built with -DCELO_MUL4K_SGN=true (round-3 open finding)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L /tmp/lib_main.so
cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_sgn.so $L
timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_seam_a.py -m gpu -q 2>&1 | tail -12
cp /tmp/lib_main.so $L
