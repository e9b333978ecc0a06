#!/bin/bash
# Instruction-mix counters of one bench workload: tools/profile_sq.sh <bench arguments>
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/sq -o sq -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/sq.json 2> $O/sq.err || { echo FAILED; tail -5 $O/sq.err; exit 1; }
echo ok
