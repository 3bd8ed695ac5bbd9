# Same-box A/B of kernel variants: [AB_FLAGS='--config 4 --robots 4096 --steps 5 --warmup 3'] ab_lib.sh <name> <name> ...  (three alternating rounds of the bench per variant)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$v.so python bench.py --seam solver $AB_FLAGS --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', round(d['value']), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4), 'err', d.get('max_grf_err_vs_osqp'))"
  done
done
