#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE / SQ passes (tools/profile_bench.sh)
# of the default bench command, the all-groups MSM table at 2^20, bench.py --config 3 / 4 / 5 and the pairing throughput tool.
# Raw CSVs land under gpurun_out/prof_<tag>/; tools/summarise_profile.py <tag> writes profiles/<tag>_summary.md + <tag>_traffic.json.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r3}
cd $ROOT
bash tools/profile_bench.sh ${R} > gpurun_out/profile_${R}.log 2>&1
bash tools/profile_bench.sh ${R}_groups "python $ROOT/tools/bench_groups.py 20" > gpurun_out/profile_${R}_groups.log 2>&1
bash tools/profile_bench.sh ${R}_cfg3 "python $ROOT/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline" > gpurun_out/profile_${R}_cfg3.log 2>&1
bash tools/profile_bench.sh ${R}_cfg4 "python $ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline" > gpurun_out/profile_${R}_cfg4.log 2>&1
bash tools/profile_bench.sh ${R}_cfg5 "python $ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline" > gpurun_out/profile_${R}_cfg5.log 2>&1
bash tools/profile_bench.sh ${R}_pairing "python $ROOT/tools/bench_pairing.py 81920" > gpurun_out/profile_${R}_pairing.log 2>&1
# keep the merge small: the per-dispatch trace CSVs are not needed once the stats exist
find gpurun_out -name "*_kernel_trace.csv" -size +8M -delete
du -sh gpurun_out/prof_${R}*
tail -2 gpurun_out/profile_${R}*.log
