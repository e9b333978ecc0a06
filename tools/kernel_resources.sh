#!/bin/bash
# Register, LDS and scratch use of every kernel in one source file of libmpc_amd (compiler remarks, no GPU needed).
#   tools/kernel_resources.sh mpc_solver.hip [extra hipcc flags]
cd "$(dirname "$0")/../mpc_amd/csrc" || exit 1
f=$1; shift
extra=""
case $f in mpc_solver.hip|mpc_nl.hip) extra="-mllvm -disable-machine-licm";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed -e 's/.*remark: [^ ]* //' | paste - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
