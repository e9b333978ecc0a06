#!/bin/bash
# HBM / L2 traffic counters of one bench workload (one counter group per pass): tools/profile_mem.sh <bench arguments>
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
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mpc%' order by dispatch_id, counter_name").fetchall()
last = {}
for d, k, n, v, dur in rows:
    last.setdefault(d, {"kernel": k[:60], "ns": dur})[n] = v
for d in sorted(last)[-2:]:
    print(d, last[d])
PYEOF
  find $O/$name -name "*_results.db" -delete
}
pass m1 "FETCH_SIZE" "$@" && pass m2 "WRITE_SIZE" "$@" && pass m3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@" && pass m4 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "$@" && echo ok
