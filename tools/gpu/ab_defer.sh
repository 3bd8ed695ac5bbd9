# Same-box A/B of the job kernel's polish schedule (MPC_POLISH_DEFER: quarters of the workgroup count below which an ADMM job's polish is left for the tail;
# 1048576 = every polish deferred, the round-3 schedule): ab_defer.sh <value> <value> ...
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    MPC_POLISH_DEFER=$v python bench.py --seam solver --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('defer $v round $round', round(d['value']), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4))"
  done
done
