# Same-box A/B of library variants at chosen horizons (solver seam, 4096 robots): HS="10 12 16 20" ab_h.sh <name> <name> ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    for H in ${HS:-10 12 16 20}; do
      C=2; [ $H = 16 ] && C=4; [ $H = 20 ] && C=5
      MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$v.so python bench.py --seam solver --config $C --horizon $H --robots 4096 --steps 5 --warmup 3 --repeats 3 --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v h=$H round $round', round(d['value']), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4), 'iters', round(d['mean_admm_iters'],2), 'solved', d['solved_fraction'])"
    done
  done
done
