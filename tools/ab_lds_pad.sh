P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
for pad in 0 8192 24000 50000 100000; do
echo "lds pad $pad"
TSDIFF_COMBO_LDS_PAD=$pad python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
