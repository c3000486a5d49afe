# timing-only ablations of the per-block launch (results are WRONG with any bit set)
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_us"])'
for a in 0 16 32 48 8 56 59; do
echo -n "ablate $a: "
TSDIFF_ABLATE=$a python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
