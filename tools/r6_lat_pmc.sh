#!/bin/bash
# where does a lone wave's time go?  SQ counters of the latency-path kernels of verify_signature (tools/bench_latency.py)
O=gpurun_out/r6_lat_pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $GRAFT_REPO_ROOT/$O/sq -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_latency.py > $GRAFT_REPO_ROOT/$O/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d $GRAFT_REPO_ROOT/$O/sq2 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_latency.py > $GRAFT_REPO_ROOT/$O/log2.txt 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_latency.py > $GRAFT_REPO_ROOT/$O/log3.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob, collections
for d in ("sq", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r6_lat_pmc/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "k377_wide" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-20:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(d, k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
for f in glob.glob("gpurun_out/r6_lat_pmc/trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k377_wide" in r["Name"]: print(r["Name"].split("(")[0][-22:], r["Calls"], r["AverageNs"])
P
tail -2 $O/log1.txt
find $O -name "*kernel_trace.csv" -delete
