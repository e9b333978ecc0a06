#!/bin/bash
# Kernel-trace summaries and HBM counter passes for the bench workloads (run on the GPU box through gpurun).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
run() { name=$1; shift; timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; exit 1; }; }
run c2_fused --workload c2 &&
run c2_perstep --workload c2 --per-step-launch --steps 300 --warmup 50 &&
run c3 --workload c3 --steps 30 --warmup 5 &&
run c4 --workload c4 --steps 100 --warmup 10 &&
run c5 --workload c5 --steps 10 --warmup 2 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --per-step-launch --steps 100 --warmup 20 > $O/pmc_fetch.json 2> $O/pmc_fetch.err &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc_write -- python3 $R/bench.py --no-cpu-baseline --per-step-launch --steps 100 --warmup 20 > $O/pmc_write.json 2> $O/pmc_write.err &&
python3 $R/tools/summarize_profiles.py $O $O/summary r01e && find $O -name "*_results.db" -delete && ls $O/summary
