#!/bin/bash
# Instruction counters of the last k_mpc dispatches of any script: tools/profile_sq_script.sh <name> <script.py> [args]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
name=$1; shift
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/$name -o $name -- python3 "$@" > $O/$name.out 2> $O/$name.err || { echo FAILED $name; tail -5 $O/$name.err; exit 1; }
python3 - <<PYEOF
import sqlite3
c = sqlite3.connect("$O/$name/${name}_results.db")
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mpc%' order by dispatch_id, counter_name").fetchall()
last = {}
for d, k, n, v, dur in rows:
    last.setdefault(d, {"kernel": k[:40], "ns": dur})[n] = v
for d in sorted(last)[-3:]:
    print(d, last[d])
PYEOF
find $O/$name -name "*_results.db" -delete
