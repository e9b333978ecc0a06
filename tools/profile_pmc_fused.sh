#!/bin/bash
# HBM counter passes over the default (fused) bench run: one pass per counter.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d $O/pmcf_$c -o pmcf_$c -- python3 $R/bench.py --no-cpu-baseline > $O/pmcf_$c.json 2> $O/pmcf_$c.err || { echo FAILED $c; tail -5 $O/pmcf_$c.err; exit 1; }
done
python3 $R/tools/summarize_profiles.py $O $O/summary r01e && python3 - <<PYEOF
import sqlite3, json
out = {}
for n, key in (("pmcf_FETCH_SIZE", "fetch_kb"), ("pmcf_WRITE_SIZE", "write_kb")):
    c = sqlite3.connect("$O/%s/%s_results.db" % (n, n))
    rows = c.execute("select dispatch_id, value, duration from counters_collection where kernel_name like '%k_mpc<%' order by dispatch_id").fetchall()
    out[key] = rows[-1][1]; out[key + "_launch_ns"] = rows[-1][2]
out["command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --no-cpu-baseline (one pass per counter; the 1000-step timed launch of k_mpc<50, fixed-model>)"
out["hbm_bytes_per_launch"] = 2 * out["fetch_kb"] * 1024 + out["write_kb"] * 1024
out["note"] = "FETCH_SIZE doubled as prescribed for gfx950 streaming reads (upper estimate for this kernel's narrow accesses); WRITE_SIZE as reported."
json.dump(out, open("$O/summary/r01e_c2_fused_hbm_traffic.json", "w"), indent=1)
print(out)
PYEOF
find $O -name "*_results.db" -delete
echo done
