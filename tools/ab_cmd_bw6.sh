python tools/bench_groups.py 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v['accumulate_ms'],3), round(v['total_ms'],3)) for k,v in d.items() if isinstance(v,dict)})"
python bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 ms', round(d['ms_per_step'],3))"
