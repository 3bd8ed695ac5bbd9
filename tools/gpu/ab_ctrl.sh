# Same-box A/B of library variants on the control-loop leg of bench.py: ab_ctrl.sh <variant> <variant> ...
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for v in "$@"; do
  MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['control_loop']; print('$v round $round ms/tick', round(c['ms_per_tick'],4), 'ticks/s', round(c['robot_ticks_per_s']), 'with resets', round(d['control_loop_with_resets']['robot_ticks_per_s']))"
done; done
