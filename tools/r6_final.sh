#!/bin/bash
# round 6, final measurement set, part 1: the world-size-8 shared-GPU dry runs, then the rocprofv3 trace + PMC passes of every configuration
# (tools/profiles_all.sh r6).  Part 2 (after tools/summarise_profile.py has written profiles/r6_*_traffic.json): python bench.py --all-configs
O=gpurun_out/r6_final; mkdir -p $O
export TMPDIR=/tmp
TAG=r6 timeout 2400 bash tools/world8.sh 3 > $O/world8.log 2>&1; tail -12 $O/world8.log
timeout 3000 bash tools/profiles_all.sh r6 > $O/profiles_all.log 2>&1; tail -5 $O/profiles_all.log
