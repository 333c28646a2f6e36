#!/bin/bash
# Round-3 open finding, localisation 2: the AUX G2 unit (k_gen_points<Fq2>, the batched MSM kernels) with the signed pass at selected sites
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L /tmp/lib_main.so
python tools/r4_gen_compare.py save /tmp/g2pts.npy 17
T='tests/test_msm_gpu.py::test_g2_plain_entry_two_to_17_and_20 tests/test_msm_gpu.py::test_g2_subgroup_entry_glv_split_device_resident'
for v in "$@"; do
  cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_$v.so $L
  echo "== variant $v"; python tools/r4_gen_compare.py cmp /tmp/g2pts.npy 17 2>&1 | tail -8
  timeout 900 python -m pytest $T -m gpu -q 2>&1 | tail -3
done
cp /tmp/lib_main.so $L
