#!/bin/bash
# A/B of prebuilt libraries on the c2 driver window and steady state: tools/ab_libs.sh lib1.so lib2.so ... (paths under mpc_amd/lib/)
# Each library is a developer build for N = 50 (MPC_DEV_HORIZONS=50); two passes over the list so that drift shows.
export MPC_DEV_HORIZONS=50
for pass in 1 2; do
  for l in "$@"; do
    MPC_AMD_LIBRARY=$PWD/mpc_amd/lib/$l timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-launches 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l pass $pass: window %.3f ms (%.1f M/s)  steady %.3f ms' % (d['roofline']['avg_launch_ms'], d['value']/1e6, d['solver']['steady_state']['launch_ms']))" || exit 1
  done
done
