cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in b16 nt2 nt2w3; do
    MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$v.so python bench.py --seam solver --config 4 --robots 4096 --steps 5 --warmup 3 --repeats 3 --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', round(d['value']), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4), 'solved', d['solved_fraction'])"
  done
done
