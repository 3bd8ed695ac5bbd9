#!/bin/bash
# Build an experimental copy of the solver library for same-box A/B runs (tools/gpu/ab_lib.sh, ab_exact.sh) and profiling builds:
#   [HLIST='X(10)'] tools/build_variant.sh NAME [extra hipcc flags...]
# -> rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_NAME.so  (select with MPC_LIB_PATH=...)
# HLIST: the planning horizons to compile (default: the product's list); 'X(10)' builds in a third of the time.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/rl-mpc-locomotion_amd/csrc/variants"
hl=()
[ -n "${HLIST:-}" ] && hl=("-DMPC_HORIZON_LIST(X)=$HLIST")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "${hl[@]}" "$@" \
  "$root/rl-mpc-locomotion_amd/csrc/mpc_batch.hip" -o "$root/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$name.so"
echo "built variants/libmpc_batch_$name.so"
