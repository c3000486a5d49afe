P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
for i in 1 2 3; do python bench.py --workload c5 --graphs 1024 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; done
for i in 1 2 3; do python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; done
