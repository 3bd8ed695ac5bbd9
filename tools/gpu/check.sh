# GPU tests, the headline bench line and the secondary configurations in short (after a kernel change).
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'solve', round(r['kernel_ms'],4), 'prep', round(r['prep_kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic MB', round((r['traffic'] or 0)/1e6,1), 'err', d.get('max_grf_err_vs_osqp'))
for k,v in d['secondary'].items(): print(' ', k, round(v['control_steps_per_s']), v.get('solve_kernel_ms', v.get('solve_kernels_ms')), v.get('prep_kernel_ms'))
print('  control loop', round(d['control_loop']['robot_ticks_per_s']), 'with resets', round(d['control_loop_with_resets']['robot_ticks_per_s']))"
