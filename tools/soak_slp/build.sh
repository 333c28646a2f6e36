#!/bin/bash
# builds celo-bls-snark-rs_amd/build/soak_slp: k_accumulate<G1_377> with and without the SLP vectorizer side by side (see main.hip)
set -e
cd "$(dirname "$0")/../.."
B=celo-bls-snark-rs_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-long-branch-factor=0"
mkdir -p $B
hipcc $F -DVARIANT=s -c tools/soak_slp/kernel.hip -o $B/soak_slp_s.o 2>/dev/null &
hipcc $F -fno-slp-vectorize -DVARIANT=n -Dcelo=celo_n -c tools/soak_slp/kernel.hip -o $B/soak_slp_n.o 2>/dev/null &
hipcc $F -c tools/soak_slp/main.hip -o $B/soak_slp_main.o &
wait
hipcc --offload-arch=gfx950 -o $B/soak_slp $B/soak_slp_main.o $B/soak_slp_s.o $B/soak_slp_n.o
