#!/bin/bash
# A/B of prebuilt libraries on another workload: tools/ab_workload.sh c4 lib1.so lib2.so ... (developer builds for N = 50)
export MPC_DEV_HORIZONS=50
w=$1; shift
for pass in 1 2; do
  for l in "$@"; do
    MPC_AMD_LIBRARY=$PWD/mpc_amd/lib/$l timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --latency-launches 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['solver']; print('$l pass $pass: %.3f M/s  %.3f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: s[k] for k in s if 'active_set' in k or k in ('mean_iterations',)})" || exit 1
  done
done
