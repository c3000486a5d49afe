P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
for pr in 0 1 0 1; do
echo "node prio $pr"
TSDIFF_NODE_PRIO=$pr python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
for pr in 0 1; do
TSDIFF_NODE_PRIO=$pr python bench.py --workload c5 --graphs 1024 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
TSDIFF_NODE_PRIO=$pr python bench.py --models 8 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
