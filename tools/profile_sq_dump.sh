#!/bin/bash
# Instruction-mix and stall counters of one bench workload, printed per dispatch of the MPC kernels (two counter passes):
#   tools/profile_sq_dump.sh <bench arguments>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
pass() {
  name=$1; shift; ctrs=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc $ctrs -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline --latency-launches 0 "$@" > $O/$name.json 2> $O/$name.err || { echo FAILED $name; tail -5 $O/$name.err; exit 1; }
  python3 - <<PYEOF
import sqlite3
c = sqlite3.connect("$O/$name/${name}_results.db")
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mpc%' or kernel_name like '%k_ipm%' or kernel_name like '%k_light%' order by dispatch_id, counter_name").fetchall()
last = {}
for d, k, n, v, dur in rows:
    last.setdefault(d, {"kernel": k[:70], "ns": dur})[n] = v
for d in sorted(last)[-3:]:
    print(d, last[d])
PYEOF
  find $O/$name -name "*_results.db" -delete
}
pass sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "$@" &&
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "$@" &&
echo ok
