#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench command.
# Writes raw output under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by tools/summarise_profile.py.
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=${2:-"python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"}
echo "$BENCH" > $OUT/command.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc --output-format csv -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc --output-format csv -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/pmc_sq -o pmc --output-format csv -- $BENCH > $OUT/pmc_sq.log 2>&1
# effective shader clock under DVFS: GRBM_GUI_ACTIVE (GPU-busy cycles) per dispatch over its duration anchors the VALU peak
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_clock -o pmc --output-format csv -- $BENCH > $OUT/pmc_clock.log 2>&1
find $OUT -name "*.csv" | head -50
tail -3 $OUT/trace.log
