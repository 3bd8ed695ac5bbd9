#!/bin/bash
# Build an experimental copy of the solver library for same-box A/B runs (tools/gpu/ab_lib.sh, ab_exact.sh) and profiling builds:
#   [HORIZONS='10 16'] tools/build_variant.sh NAME [extra hipcc flags...]
# -> rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_NAME.so  (select with MPC_LIB_PATH=...)
# HORIZONS: the planning horizons to compile (default '10 16 20'); one translation unit per horizon, built in parallel (csrc/Makefile).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/rl-mpc-locomotion_amd/csrc/variants"
make -s -j8 -C "$root/rl-mpc-locomotion_amd/csrc" HORIZONS="${HORIZONS:-10 16 20}" OUT="variants/libmpc_batch_$name.so" OBJ="_obj_$name" EXTRA="$*" 2>&1 | grep -E "error|Error" || true
ls -la "$root/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$name.so"
