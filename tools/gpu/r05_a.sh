# round 5, first GPU pass: the whole -m gpu suite (observed parity figures printed), policy kernel A/B (512 vs 256 threads per workgroup), the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r05a_gputest.log 2>&1; tail -3 gpurun_out/r05a_gputest.log
grep -E "compared|bridge_|estimator samples|max torque error|FAILED|Error" gpurun_out/r05a_gputest.log | head -60
for v in "" variants/libmpc_batch_pol256.so; do
  for round in 1 2; do
    MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/${v:-libmpc_batch.so} python -c "
import bench, torch, json
r = bench.policy_leg(4096, torch.device('cuda:0'))
print('policy', '${v:-shipped(512)}', {k: (round(x, 5) if isinstance(x, float) else x) for k, x in r.items() if k != 'note' and k != 'what'})" 2>&1 | tail -1
  done
done
python bench.py --no-cpu-baseline > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r05a_bench.json')); r = d['roofline']
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'solve', round(r['kernel_ms'], 4), 'prep', round(r['prep_kernel_ms'], 4), 'frac', round(r['frac'], 4))
for k, v in d['secondary'].items(): print(' ', k, round(v['control_steps_per_s']), v.get('ms_per_step'), v.get('solve_kernel_ms', v.get('solve_kernels_ms')), v.get('prep_kernel_ms'))
print('  control loop', round(d['control_loop']['robot_ticks_per_s']), 'with resets', round(d['control_loop_with_resets']['robot_ticks_per_s']), 'policy', d['policy'])
PY
