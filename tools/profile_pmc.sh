# HBM traffic counters, each in its own pass (rocprofv3 PMC slots), kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_summary.py $(ls /tmp/pmc_f/*/*results.db /tmp/pmc_f/*results.db 2>/dev/null | head -1) $(ls /tmp/pmc_w/*/*results.db /tmp/pmc_w/*results.db 2>/dev/null | head -1) gpurun_out/r01_pmc_traffic | head -12
