#!/bin/bash
# Build the current tree as a kernel variant for same-box A/B runs (tools/gpu/ab_lib.sh):
#   [HLIST='X(16)'] tools/ab_build.sh <name> [extra hipcc flags]   ->  rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_<name>.so   (horizon 10 only unless HLIST says otherwise)
set -e
name=$1; shift
cd "$(dirname "$0")/../rl-mpc-locomotion_amd/csrc"
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "-DMPC_HORIZON_LIST(X)=${HLIST:-X(10)}" "$@" mpc_batch.hip -o variants/libmpc_batch_$name.so
echo built variants/libmpc_batch_$name.so
