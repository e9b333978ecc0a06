#!/bin/bash
# Instruction-fetch counters of the c2 driver window (those of the candidates this rocprofv3 knows):
#   tools/profile_ifetch.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
timeout -k 10 60 rocprofv3 -L > $O/counters.txt 2>&1
have=""
for c in SQ_IFETCH SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_WAVE_CYCLES; do
  grep -qw "$c" $O/counters.txt && have="$have $c"
done
echo "counters:$have"
name=ifetch
timeout -s KILL 170 rocprofv3 --pmc $have -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline --latency-launches 0 --workload c2 --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err || { echo FAILED $name; tail -5 $O/$name.err; exit 1; }
python3 - <<PYEOF
import sqlite3
c = sqlite3.connect("$O/$name/${name}_results.db")
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mpc%' order by dispatch_id, counter_name").fetchall()
last = {}
for d, k, n, v, dur in rows:
    last.setdefault(d, {"kernel": k[:40], "ns": dur})[n] = v
for d in sorted(last)[-4:]:
    print(d, last[d])
PYEOF
find $O/$name -name "*_results.db" -delete
