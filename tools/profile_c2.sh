cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01_bench_c2.json 2> gpurun_out/c2_err.log
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/c2_prof.log 2>&1
python3 tools/rocpd_stats.py $(ls /tmp/prof_c2/*/*results.db /tmp/prof_c2/*results.db 2>/dev/null | head -1) > gpurun_out/r01_kernel_stats_c2.md
python bench.py --workload train --graphs 200 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r01_bench_train.json 2>> gpurun_out/c2_err.log
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- python3 bench.py --workload train --graphs 200 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/train_prof.log 2>&1
python3 tools/rocpd_stats.py $(ls /tmp/prof_train/*/*results.db /tmp/prof_train/*results.db 2>/dev/null | head -1) > gpurun_out/r01_kernel_stats_train.md
cut -c1-400 gpurun_out/r01_bench_c2.json; head -8 gpurun_out/r01_kernel_stats_c2.md; cut -c1-200 gpurun_out/r01_bench_train.json; head -8 gpurun_out/r01_kernel_stats_train.md
