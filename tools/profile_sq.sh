#!/bin/bash
# Instruction-mix counters of the per-step kernel in steady state (one launch per step).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/sq -o sq -- python3 $R/bench.py --no-cpu-baseline --workload c4 --steps 100 --warmup 20 > $O/sq.json 2> $O/sq.err || { echo FAILED; tail -5 $O/sq.err; exit 1; }
echo ok
