P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
for ps in 0 2 4 6 9 12; do
echo "phase sleep $ps"
TSDIFF_PHASE_SLEEP=$ps python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
