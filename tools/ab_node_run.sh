P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
for nr in 1 2 4 8; do
echo "node run $nr"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --node-run $nr 2>&1 | tail -1 | python -c "$P"
python bench.py --workload c5 --graphs 1024 --steps 10 --warmup 2 --no-cpu-baseline --node-run $nr 2>&1 | tail -1 | python -c "$P"
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
