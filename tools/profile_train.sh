cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload train --graphs 200 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r01_bench_train.json 2> gpurun_out/train_err.log
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- python3 bench.py --workload train --graphs 200 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/train_prof.log 2>&1
python3 tools/rocpd_stats.py $(ls /tmp/prof_train/*/*results.db /tmp/prof_train/*results.db 2>/dev/null | head -1) > gpurun_out/r01_kernel_stats_train.md
cat gpurun_out/r01_bench_train.json | cut -c1-200
head -16 gpurun_out/r01_kernel_stats_train.md; tail -1 gpurun_out/r01_kernel_stats_train.md
