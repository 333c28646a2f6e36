#!/bin/bash
# Round-3 open finding, localisation: the big-path G2 unit built with the signed Fq2 pass at selected call sites only
# (fp2.h CELO_MUL4K_SGN_SITES: 4 = xyzz_madd - the accumulate kernel; 19 = xyzz_dbl_affine + xyzz_dbl + xyzz_add - the reduction kernels)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=celo-bls-snark-rs_amd/build/libcelo_bls_amd.so
cp $L /tmp/lib_main.so
T='tests/test_msm_gpu.py::test_g2_plain_entry_two_to_17_and_20 tests/test_msm_gpu.py::test_g2_subgroup_entry_glv_split_device_resident tests/test_msm_gpu.py::test_bw6_761_g2_two_to_17'
echo "== main library"; timeout 900 python -m pytest $T -m gpu -q 2>&1 | tail -4
for v in "$@"; do
  cp celo-bls-snark-rs_amd/build_ab/libcelo_bls_amd_$v.so $L
  echo "== variant $v"; timeout 900 python -m pytest $T -m gpu -q 2>&1 | tail -9
done
cp /tmp/lib_main.so $L
