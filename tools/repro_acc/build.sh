#!/bin/bash
# builds celo-bls-snark-rs_amd/build/repro_acc[_TAG]: the unsigned and the signed-madd k_accumulate<G2_377> side by side (see main.hip).
# usage: build.sh [TAG "extra flags for the SIGNED kernel's compilation"]
#   build.sh                               signed xyzz_madd pass, per-lane zero test (the library's): differs from the unsigned kernel about once
#                                          per 10^6 additions (seen through the library: tools/r4_sgn_sites.sh)
#   build.sh uni "-DCELO_ZERO_UNIFORM"     signed pass + wave-uniform zero test: ~0.7 % of the waves wrong, deterministically (the compact form)
#   SITES=0 build.sh same ""               two unsigned copies: must agree (and agree with the host replay)
set -e
cd "$(dirname "$0")/../.."
B=celo-bls-snark-rs_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
TAG=${1:+_$1}
[ -f $B/repro_acc_u.o ] || hipcc $F -DVARIANT=u -c tools/repro_acc/kernel.hip -o $B/repro_acc_u.o &
[ -f $B/repro_acc_main.o ] || hipcc $F -c tools/repro_acc/main.hip -o $B/repro_acc_main.o &
hipcc $F $2 -DVARIANT=s -Dcelo=celo_s -DCELO_MUL4K_SGN_SITES=${SITES:-4} -c tools/repro_acc/kernel.hip -o $B/repro_acc_s$TAG.o &
wait
hipcc --offload-arch=gfx950 -o $B/repro_acc$TAG $B/repro_acc_main.o $B/repro_acc_u.o $B/repro_acc_s$TAG.o $B/host_ifma.o $B/host_cpu.o
