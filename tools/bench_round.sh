#!/bin/bash
# The bench lines of a round in one gpurun call:   gpurun -- 'bash tools/bench_round.sh r03'
# Writes gpurun_out/<tag>_bench_*.json (one JSON line each) and gpurun_out/<tag>_parity_report.md.  Copy what is to
# be judged to profiles/.  Run tools/profile_round.sh first when the line's `traffic` / `mfma_busy` fields are to
# come from this round's counter passes (bench.py reads profiles/<tag>_pmc_traffic*.json and <tag>_mfma_busy.json).
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err      # the driver's command
python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c2_1000.json 2> /dev/null
python3 bench.py --steps 5000 --warmup 50 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c2_5000.json 2> /dev/null   # the real job length
python3 bench.py --workload c5 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c5.json 2> /dev/null
python3 bench.py --models 8 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/${TAG}_bench_ensemble8.json 2> /dev/null
python3 bench.py --models 8 --graphs 300 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c3_unit.json 2> /dev/null  # configs[2]: one GPU's share
python3 bench.py --workload train --steps 20 --warmup 5 > $O/${TAG}_bench_train.json 2> /dev/null
python3 bench.py --workload train --steps 20 --warmup 5 --no-prefetch > $O/${TAG}_bench_train_no_prefetch.json 2> /dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload train --steps 40 --warmup 8 > $O/${TAG}_bench_train_torchrun_3range.json 2> /dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --workload train --steps 40 --warmup 8 --single-range-reduce > $O/${TAG}_bench_train_torchrun_1range.json 2> /dev/null
python3 tests/tools/parity_report.py > $O/${TAG}_parity_report.md 2> $O/${TAG}_parity_report.err
for f in $O/${TAG}_bench_*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["steps"], d["ms_per_step"], d["value"], d["unit"], (d.get("roofline") or {}).get("frac"))
PY
done
tail -3 $O/${TAG}_bench_driver_flags.err
