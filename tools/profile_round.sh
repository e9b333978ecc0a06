#!/bin/bash
# Kernel-trace summaries and HBM counter passes for the bench workloads (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>        e.g. r02
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
run() { name=$1; shift; timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; exit 1; }; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
pmc() { name=$1; shift; ctr=$1; shift; timeout -s KILL 170 rocprofv3 --pmc $ctr -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline --latency-launches 0 "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; exit 1; }; }
run c2_driver --workload c2 --steps 20 --warmup 5 &&
run c2_default --workload c2 &&
run c2_perstep --workload c2 --per-step-launch --steps 300 --warmup 50 &&
run c3 --workload c3 &&
run c4 --workload c4 &&
run c5 --workload c5 &&
pmc pmcf_c2_driver FETCH_SIZE --workload c2 --steps 20 --warmup 5 &&
pmc pmcw_c2_driver WRITE_SIZE --workload c2 --steps 20 --warmup 5 &&
pmc pmcf_c2_default FETCH_SIZE --workload c2 &&
pmc pmcw_c2_default WRITE_SIZE --workload c2 &&
pmc pmcf_c3 FETCH_SIZE --workload c3 &&
pmc pmcw_c3 WRITE_SIZE --workload c3 &&
pmc pmcf_c4 FETCH_SIZE --workload c4 &&
pmc pmcw_c4 WRITE_SIZE --workload c4 &&
pmc pmcf_c5 FETCH_SIZE --workload c5 &&
pmc pmcw_c5 WRITE_SIZE --workload c5 &&
pmc sq_c2_driver "$SQ" --workload c2 --steps 20 --warmup 5 &&
pmc sq_c3 "$SQ" --workload c3 &&
pmc sq_c4 "$SQ" --workload c4 &&
pmc sq_c5 "$SQ" --workload c5 &&
python3 $R/tools/summarize_profiles.py $O $O/summary $TAG && find $O -name "*_results.db" -delete && ls $O/summary
