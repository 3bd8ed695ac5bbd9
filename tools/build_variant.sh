#!/bin/bash
# Build an experimental copy of the solver library: tools/build_variant.sh NAME [extra hipcc flags...]
# -> rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_NAME.so  (select with MPC_LIB_PATH=...)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/rl-mpc-locomotion_amd/csrc/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" \
  "$root/rl-mpc-locomotion_amd/csrc/mpc_batch.hip" -o "$root/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$name.so"
