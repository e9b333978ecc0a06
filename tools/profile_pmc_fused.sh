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
echo done
