cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d /tmp/pmc_clk -o clk -- python3 bench.py $@ --no-cpu-baseline > gpurun_out/pmc_clk.log 2>&1
python3 tools/pmc_kernel_table.py $(ls /tmp/pmc_clk/*/*results.db /tmp/pmc_clk/*results.db 2>/dev/null | head -1) "$FILTER"
: <<PY
import sqlite3, glob
db = (glob.glob('/tmp/pmc_clk/*/*results.db') + glob.glob('/tmp/pmc_clk/*results.db'))[0]
c = sqlite3.connect(db)
print([r[0] for r in c.execute("select name from sqlite_master where type='table'").fetchall()][:40])
PY
