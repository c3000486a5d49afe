P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for pre in 0 1; do
echo "pair pre $pre"
TSDIFF_PAIR_PRE=$pre python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
TSDIFF_PAIR_PRE=$pre python bench.py --workload c5 --graphs 1024 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
TSDIFF_PAIR_PRE=$pre python bench.py --models 8 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
