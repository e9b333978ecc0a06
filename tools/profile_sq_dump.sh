#!/bin/bash
# Instruction-mix counters of one bench workload, printed per dispatch of the MPC kernels:
#   tools/profile_sq_dump.sh <bench arguments>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/sq -o sq -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/sq.json 2> $O/sq.err || { echo FAILED; tail -5 $O/sq.err; exit 1; }
python3 - <<PYEOF
import sqlite3
c = sqlite3.connect("$O/sq/sq_results.db")
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mpc%' order by dispatch_id, counter_name").fetchall()
last = {}
for d, k, n, v, dur in rows:
    last.setdefault(d, {"kernel": k[:60], "ns": dur})[n] = v
for d in sorted(last)[-3:]:
    print(d, last[d])
PYEOF
find $O -name "*_results.db" -delete
echo ok
