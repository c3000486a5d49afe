# SQ issue/stall counters + MFMA-busy of the hot kernels (one PMC pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES -d /tmp/pmc_sq -o sq -- python3 bench.py $@ --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
python3 tools/pmc_kernel_table.py $(ls /tmp/pmc_sq/*/*results.db /tmp/pmc_sq/*results.db 2>/dev/null | head -1) "$FILTER"
