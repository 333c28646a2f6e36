#!/bin/bash
# instruction-side counters of the throughput pairing kernels (one pass per counter group)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=$ROOT/gpurun_out/r4_pair_pmc
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*" | sort -u > $OUT/avail.txt
BENCH="python tools/bench_pairing.py 81920"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- bash -c "cd $ROOT && $BENCH" > $OUT/g$i.log 2>&1 )
done
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r4_pair_pmc")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as o:
    for k in tot:
        if "miller" in k or "final" in k or "exp" in k:
            o.write(k + "\n")
            for c in sorted(tot[k]): o.write("  %-32s %.4g per launch (%d launches)\n" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
print(open(out + "/summary.txt").read())
PY
