#!/bin/bash
# single-caller latency: the reference's criterion shapes through Seam A, then the phase split of verify_signature calls
O=gpurun_out/r6_latency; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/bench_criterion_shapes.py 300 20 > $O/criterion_shapes.json 2> $O/criterion_shapes.err; echo "criterion rc=$?"; tail -c 1500 $O/criterion_shapes.json
CELO_AMD_LOG=1 timeout 300 python tools/bench_criterion_shapes.py 6 2 > /dev/null 2> $O/phases.txt; grep -c "verify" $O/phases.txt; grep "verify_with\|verify_hash" $O/phases.txt | tail -24
CELO_CRH_DEVICE_LIMBS=1 CELO_AMD_LOG=1 timeout 300 python tools/bench_criterion_shapes.py 6 2 > /dev/null 2> $O/phases_device_limbs_crh.txt; grep "verify_with" $O/phases_device_limbs_crh.txt | tail -6
