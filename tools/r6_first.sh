#!/bin/bash
# round 6, first GPU call: the long-branch reproducer (p vs f, then the h build under its watchdog), the SLP differential soak, the witness-like
# BW6-761 MSM that hung in round 5, then the whole GPU suite under its per-test wall-clock bounds.
O=gpurun_out/r6_first; mkdir -p $O
B=celo-bls-snark-rs_amd/build
export TMPDIR=/tmp
timeout 300 $B/repro_combine > $O/repro_combine.txt 2>&1; echo "repro_combine rc=$?" | tee -a $O/repro_combine.txt
timeout 600 $B/soak_slp > $O/soak_slp.txt 2>&1; echo "soak_slp rc=$?" | tee -a $O/soak_slp.txt
timeout 300 python tools/dbg_witness.py bw6_761_g1 18 > $O/dbg_witness.txt 2>&1; echo "dbg_witness rc=$?" | tee -a $O/dbg_witness.txt
timeout 300 python tools/dbg_witness.py bw6_761_g1 21 >> $O/dbg_witness.txt 2>&1; echo "dbg_witness 21 rc=$?" | tee -a $O/dbg_witness.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; tail -c 600 $O/bench_cfg2.json
# last: the hazard build (expected: HUNG after 20 s, exit 3; the watchdog ends the process, the kernel dies with it)
timeout -s KILL 60 $B/repro_combine hang > $O/repro_combine_hang.txt 2>&1; echo "repro_combine hang rc=$?" | tee -a $O/repro_combine_hang.txt
tail -3 $O/repro_combine_hang.txt
timeout 60 python -c "
import torch; x=torch.ones(4,device='cuda'); print('device alive after the hang:', float(x.sum()))" 2>&1 | tail -1
