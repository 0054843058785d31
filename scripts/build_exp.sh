#!/bin/bash
# Experimental build of libmpcx with extra -D flags: scripts/build_exp.sh <name> [flags...]
# -> model_predictive_control_amd/lib/exp/libmpcx_<name>.so (selected at run time with MPCX_LIB)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/model_predictive_control_amd/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" \
  -I $ROOT/include -I $ROOT/model_predictive_control_amd/csrc \
  -o $ROOT/model_predictive_control_amd/lib/exp/libmpcx_$NAME.so $ROOT/model_predictive_control_amd/csrc/mpcx_kernels.hip 2>&1 \
  | grep -A8 "Function Name: ${MPCX_EXP_KERNEL:-_ZN4mpcx13k_solve_pair5ILi1ELb0}" | grep -E "VGPRs:|AGPRs|Scratch|Occupancy" | sed 's/.*remark: *//' | tr '\n' ' '
echo
