python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for l in 0 1; do echo layout $l; TSDIFF_LINEAR_LAYOUT=$l python bench.py --workload train --graphs 200 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
