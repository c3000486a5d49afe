cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload c5 --graphs 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01_bench_c5.json 2> gpurun_out/rest_err.log
python bench.py --models 8 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r01_bench_c2_ensemble8.json 2>> gpurun_out/rest_err.log
rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 -- python3 bench.py --workload c5 --graphs 1024 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/c5_prof.log 2>&1
python3 tools/rocpd_stats.py $(ls /tmp/prof_c5/*/*results.db /tmp/prof_c5/*results.db 2>/dev/null | head -1) > gpurun_out/r01_kernel_stats_c5.md
python tools/parity_report.py > gpurun_out/r01_parity_report.md 2>> gpurun_out/rest_err.log
cut -c1-300 gpurun_out/r01_bench_c5.json; cut -c1-300 gpurun_out/r01_bench_c2_ensemble8.json; head -8 gpurun_out/r01_kernel_stats_c5.md; tail -12 gpurun_out/r01_parity_report.md
