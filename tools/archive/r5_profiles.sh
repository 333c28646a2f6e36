#!/bin/bash
# Round-5 rocprofv3 sets (on the GPU box, via gpurun): the r4 list under the tag r5 + the host-pointer pipeline (k_accumulate_chunk).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/archive/r4_profiles.sh r5
bash tools/profile_bench.sh r5_hostptr "python $ROOT/tools/bench_host_pointer.py --chunks 4 --reps 5" > gpurun_out/profile_r5_hostptr.log 2>&1
find gpurun_out -name "*_kernel_trace.csv" -size +8M -delete
du -sh gpurun_out/prof_r5*
