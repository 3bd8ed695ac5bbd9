cd $GRAFT_REPO_ROOT
run() { # name, jobs
  MPC_SOLVE_JOBS=$2 MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$1.so python bench.py --seam solver --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 jobs=$2', round(d['value']), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4))"
}
for round in 1 2; do
  run base10 1024; run w2 2048; run w2 1024; run w2 1536
done
